// libnbmf_hip — NBMF-MM multiplicative-update inner loop for MI355X (gfx950 / CDNA4).
//
// What runs here (reference: siddC/nbmf_mm, src/nbmf_mm/_solver.py:5-59 and :143-175):
//   H-pass   one fused sweep over Y:  Theta = W^T H (f64 MFMA) -> ratios -> P1 = W R1, P2 = W R2
//            (f64 MFMA, Theta's accumulator registers are the B operands) + log-likelihood of the
//            CURRENT factors (= the loss of the previous iteration, _solver.py:148-155).
//   H-update k x n Beta-MAP update (_solver.py:42-47) + Beta log-prior sums (:158-159).
//   W-pass   one fused sweep over Y^T: Theta' = H'^T W -> ratios -> Q = H'(S1-S2)^T + 1 (x) sum S2
//            (algebraically _solver.py:53; one back-product instead of two).
//   W-update k x m multiplicative step, /n, column renormalise (_solver.py:53-57) or the Duchi
//            extension (README.md:27-35).
//   finalize loss assembly and the relative-change stop rule on device (_solver.py:162-174).
//
// Layout in HBM: Y is packed once into MFMA accumulator order, 16x16 tiles, one
// byte per entry on the binary path ({ym, zero-observed, valid} bits) or one double per entry,
// in two sweep orders (strip-major for each pass) so every wave streams contiguous memory.
// Factors are kept in natural [k][len] order plus two operand-ordered copies ("T": Theta operand,
// "G": gradient/back-product operand) so LDS staging is a linear copy and ds_reads are lane-linear.
//
// No BLAS, no PyTorch: plain HIP + (optionally, loaded at run time) RCCL.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/nbmf_hip.h"

// Build switches.  Three, all for measurement or self-test builds; every experiment that was measured and rejected
// (two strips per wave, rotating priorities, even placement, the replicated logarithm table, register staging, IEEE /
// residual division, inline-assembly selects, 32 KiB stages, ...) lives in docs/HISTORY.md with the commit that last
// contained it, not here.
#ifndef NBMF_NO_MFMA
#define NBMF_NO_MFMA 0   // 1 = measurement build: the sweeps without their MFMAs (see NBMF_MFMA in nbmf_pass_kernel.inc)
#endif
#ifndef NBMF_HAZARD_SEED
#define NBMF_HAZARD_SEED 0   // 1 = compile a kernel with a deliberate MFMA-after-inline-assembly hazard (build self-test only)
#endif

namespace {

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;
// which engine served the fits of this process (nbmf_engine_stats): fits the single-launch engine ran to the end,
// persistent launches that gave up, qualifying fits it declined because the context's kernel had given up before, and
// nbmf_run calls served by the launches
std::atomic<long long> g_engine_persistent_served{0}, g_engine_persistent_aborted{0}, g_engine_persistent_declined{0},
    g_engine_launches_served{0};

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  // A failed runtime call also leaves its code in the thread's "last error", which stays until somebody reads it: the
  // next launch check (HIPCHK(hipGetLastError())) of this thread -- in this context or any other -- would report it as its
  // own (an upload that ran out of memory made the NEXT, small, fit fail with "out of memory": tests/test_gpu_lifecycle.py).
  // Reading it here resets it.
  if (code == NBMF_ERR_HIP) (void)hipGetLastError();
  return code;
}

#define HIPCHK(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(NBMF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,  \
                  __LINE__);                                                                      \
  } while (0)

inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

typedef double d4 __attribute__((ext_vector_type(4)));

enum { DATA_BIN = 0, DATA_F64 = 1, DATA_F64M = 2 };
enum { MODE_H = 0, MODE_W = 1, MODE_L = 2, MODE_T = 3 };

// code bits of the binary path
enum : unsigned { CB_YM = 1u, CB_ZOBS = 2u, CB_VALID = 4u };

constexpr int SLICE_K = 128;      // n_components beyond this run as slices of SLICE_K components
constexpr int PAD = 128;          // m and n are padded to multiples of 128 (8 row blocks, 8 strips)
constexpr int WG_WAVES = 4;   // waves (= column strips) per workgroup of the pass kernel; measured: 2-wave workgroups run 30 % slower (125 vs 179 it/s at c3)
constexpr size_t PASS_SLACK = 65536;   // bytes allocated behind every image the sweeps stream: their prefetch runs one stage (<= 32 KiB) past the last block
constexpr int STAGE_BYTES = 32768; // one LDS stage: NB row blocks x (T + G operand images); two stages per workgroup

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
// Reciprocal of d in [eps, 1+eps]: hardware seed + two Newton steps (<= 1 ulp; no scaling needed in
// this range).  Stands in for the IEEE divides of _solver.py:42-43,53 on the binary path, where the
// numerator is exactly 0 or 1.
__device__ __forceinline__ double rcp_nr(double d) {
  // seed, then one cubically convergent step: r1 = r0 (1 + e + e^2), e = 1 - d r0  (3 FMAs;
  // |e| <= 2^-20 for v_rcp_f64, so the truncation term e^3 is far below 1 ulp)
  double r = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, r, 1.0);
  const double p = __builtin_fma(e, e, e);
  return __builtin_fma(r, p, r);
}

// double from its two 32-bit halves / back (bit-level selects cost one VALU op per half)
__device__ __forceinline__ double mk_double(uint32_t lo, uint32_t hi) {
  return __hiloint2double((int)hi, (int)lo);
}

// Natural logarithm for the general (real-valued / weighted) path, where two logarithms per entry feed the
// loss and every VALU instruction adds to the f64 MFMA time: a 256-entry table in LDS plus a short series,
// 13 VALU instructions and one ds_read_b128 (the library routine: ~75; the fdlibm-style routine this replaces: ~35).
//   x = 2^e * m, m in [0.5, 1);  bin j = top 8 fraction bits of m;  table[j] = {c_j, -log(c_j)}, c_j = 1/(bin centre)
//   r = m c_j - 1 (one FMA, |r| <= 2^-9);  log x = e ln2 - log c_j + log1p(r),  log1p to r^5/5 (next term 2^-56.6)
// Absolute error <= 2.5e-16 + 1 ulp of the result (the table value and e ln2 are each rounded once): sums of
// millions of such terms of size ~0.5 keep the loss well inside its 1e-10 tolerance.  GUARD: arguments that are
// not positive normal numbers (1 - Theta + eps <= 0 needs factors off the simplex; eps is at least the smallest
// normal number) take the library routine: same NaN / -inf as NumPy.
// max(x, 0) with a NaN x giving 0 (v_max_f64 returns the operand that is a number): one instruction, where the
// compiler's fmax adds a canonicalising one in front for a value it has loaded from memory.
__device__ __forceinline__ double max0(double x) {
  double r;
  __asm__("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x));
  return r;
}
constexpr int LOG_TABLE_BYTES = 256 * 16;
// (TB index bits: 8 -> 256 entries and the series to r^5/5; 10 -> 1024 entries, |r| <= 2^-11, and the series to r^4/4 --
//  the next term is below 2^-57 -- one FMA fewer per logarithm where the table is there for the taking)
template <int TB>
__device__ __forceinline__ double2 log_table_entry(int j) {
  const double c = 1.0 / (0.5 + ((double)j + 0.5) * (1.0 / (double)(2 << TB)));
  return double2{c, -log(c)};
}
__device__ __forceinline__ void log_table_fill(double2* tab /* LDS, 256 entries; threads 0..255 */) {
  if (threadIdx.x < 256) tab[threadIdx.x] = log_table_entry<8>(threadIdx.x);
}
// (LDS_ABS: `tab` is not a pointer but the table's byte address in LDS, known at compile time -- the caller's kernel has
//  no static LDS, so its dynamic region starts at 0 -- and becomes the gather's immediate offset; through a pointer
//  derived from the dynamic region's symbol hipcc spends a vector add per gather on a base it cannot fold)
template <bool GUARD, int TB = 8, bool LDS_ABS = false, typename TAB = const double2*>
__device__ __forceinline__ double log_tab(double x, TAB tab, double c3 = 0.33333333333333331483, double c4 = -0.25) {
  static_assert(TB == 8 || TB == 10, "256 or 1024 entries");
  if (GUARD && __builtin_expect(!__builtin_amdgcn_class(x, 0x100), 0)) return log(x);   // not a positive normal number
  const uint32_t hi = (uint32_t)__double2hiint(x);
  const double m = __builtin_amdgcn_frexp_mant(x);   // in [0.5, 1); a NaN stays a NaN and so reaches the result
  const uint32_t off = (hi >> (16 - TB)) & (((1u << TB) - 1u) << 4);   // byte offset of the entry: two instructions
  double2 e;
  if constexpr (LDS_ABS) {
    typedef double d2raw __attribute__((ext_vector_type(2)));
    const d2raw v = *(const __attribute__((address_space(3))) d2raw*)(uintptr_t)(off + (uint32_t)tab);
    e = double2{v[0], v[1]};
  } else {
    e = *(const double2*)((const char*)tab + off);
  }
  const double r = __builtin_fma(m, e.x, -1.0);
  double p;
  if (TB == 8) {
    p = __builtin_fma(r, 0.2, -0.25);
    p = __builtin_fma(r, p, c3);
  } else {
    p = __builtin_fma(r, c4, c3);   // (callers in a hot loop hand both constants over in registers: two literals do not fit one instruction)
  }
  p = __builtin_fma(r, p, -0.5);
  p = __builtin_fma(r, p, 1.0);
  const double k = (double)__builtin_amdgcn_frexp_exp(x);
  return __builtin_fma(r, p, __builtin_fma(k, 0.69314718055994530942, e.y));
}
// The 1024-entry table, once per device and process in memory (a workgroup of the general path's sweeps copies it
// into LDS: 16 KiB from L2 instead of 1024 library logarithms).
__global__ __launch_bounds__(256) void log_table_kernel(double2* tab) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  tab[j] = log_table_entry<10>(j);
}

// log(prod) + pexp ln 2 for the running product of a sweep's likelihood (one call per lane and sweep), without a trip to
// memory (the library's logarithm reads tables from constant memory; in a workgroup's epilogue that is a chain of cold
// misses).  Mantissa m in [1/sqrt 2, sqrt 2), s = (m - 1) / (m + 1), log m = 2 s (1 + s^2/3 + ... + s^18/19): |s| <= 0.172,
// the next term is below 2^-55.  Absolute error <= 2.5e-16 + 1 ulp of the result (op 7 of nbmf_selftest_unary).  Zero,
// NaN, infinities and negative products take the library routine: NumPy's -inf / NaN.  (Built in round 4 on the suspicion
// that the H sweep's long epilogue -- 9 us against the W sweep's 2.7 at configs[1] -- was the library call; the finer trace
// says 5 us of it are the workgroup's waves waiting for each other at the loss block's barrier, and the sweep's time did
// not change.  Kept: it is deterministic, self-contained and no slower.)
__device__ __forceinline__ double log_of_product(double prod, int pexp) {
  const double m = __builtin_amdgcn_frexp_mant(prod);
  if (__builtin_expect(!(m >= 0.5 && m < 1.0), 0))   // (zero: m = 0; negative: m < 0; NaN and infinity: the comparisons fail)
    return log(prod) + (double)pexp * 0.6931471805599453094;
  int e = __builtin_amdgcn_frexp_exp(prod) + pexp;
  const bool low = m < 0.70710678118654752440;
  const double m2 = low ? m + m : m;
  e -= low ? 1 : 0;
  const double s = (m2 - 1.0) * rcp_nr(m2 + 1.0);
  const double w = s * s;
  double p = __builtin_fma(w, 1.0 / 19.0, 1.0 / 17.0);
  p = __builtin_fma(w, p, 1.0 / 15.0);
  p = __builtin_fma(w, p, 1.0 / 13.0);
  p = __builtin_fma(w, p, 1.0 / 11.0);
  p = __builtin_fma(w, p, 1.0 / 9.0);
  p = __builtin_fma(w, p, 1.0 / 7.0);
  p = __builtin_fma(w, p, 0.2);
  p = __builtin_fma(w, p, 0.33333333333333331483);
  p = __builtin_fma(w, p, 1.0);
  return __builtin_fma(s + s, p, (double)e * 0.69314718055994530942);
}

// Per-lane selects on a 64-bit wave mask, written out because the pass kernels are VALU-issue bound next to
// their MFMAs: a byte test is ONE SDWA compare (hipcc otherwise emits v_and + v_cmp), a double select is two
// v_cndmask_b32 (with the zero / negated alternative folded into the operand).
// CAUTION for callers: hipcc does not know that these statements are vector instructions, so it does not insert the
// two wait states gfx950 needs between a vector write of a register and an MFMA that reads it.  A select's result
// must therefore not be the operand of the very next MFMA: the pass kernel has the operand loads of the product
// loop and a scheduling barrier in between, the single-launch kernel an explicit s_nop (see small_tile).
typedef unsigned long long lanemask_t;
// The four byte tests of a code word in ONE statement: LLVM's hazard table for this chip asks for two wait states
// between a vector instruction writing an SGPR pair and a vector instruction reading it; hipcc cannot see into the
// assembly, and it does move a lone compare right in front of its select.  Kept together, the compares are followed
// by the selects' own preparation (1 - Theta, ...), and tools/check_asm_mfma_hazard.py --sgpr verifies the distance
// in every build (tests/test_isa_hazards.py).  (With an s_nop 1 inside the statement instead: -0.8 % on the bench.)
__device__ __forceinline__ void byte_set4(uint32_t x, lanemask_t (&m)[4]) {
  const uint32_t zero = 0u;
  __asm__("v_cmp_ne_u32_sdwa %0, %4, %5 src0_sel:BYTE_0 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %1, %4, %5 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %2, %4, %5 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %3, %4, %5 src0_sel:BYTE_3 src1_sel:DWORD"
          : "=&s"(m[0]), "=&s"(m[1]), "=&s"(m[2]), "=&s"(m[3])
          : "v"(x), "v"(zero));
}
// The eight byte tests of a W-sweep tile (observed ones, then observed zeros) in ONE statement.  Distance by
// construction: an observed-one mask is followed by at least the four observed-zero compares; an observed-zero mask
// is first read by a select whose other operand is the reciprocal, which depends (through d = select(ym, ...) + eps) on
// selects that come after this whole statement.
__device__ __forceinline__ void byte_set8(uint32_t x, uint32_t y, lanemask_t (&m)[4], lanemask_t (&n)[4]) {
  const uint32_t zero = 0u;
  __asm__("v_cmp_ne_u32_sdwa %0, %8, %10 src0_sel:BYTE_0 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %1, %8, %10 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %2, %8, %10 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %3, %8, %10 src0_sel:BYTE_3 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %4, %9, %10 src0_sel:BYTE_0 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %5, %9, %10 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %6, %9, %10 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
          "v_cmp_ne_u32_sdwa %7, %9, %10 src0_sel:BYTE_3 src1_sel:DWORD"
          : "=&s"(m[0]), "=&s"(m[1]), "=&s"(m[2]), "=&s"(m[3]), "=&s"(n[0]), "=&s"(n[1]), "=&s"(n[2]), "=&s"(n[3])
          : "v"(x), "v"(y), "v"(zero));
}
__device__ __forceinline__ lanemask_t byte_set(uint32_t x, int byte) {   // one of them (callers: keep it two instructions away from its select)
  lanemask_t m;
  const uint32_t zero = 0u;
  switch (byte) {
    case 0: __asm__("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(m) : "v"(x), "v"(zero)); break;
    case 1: __asm__("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD" : "=s"(m) : "v"(x), "v"(zero)); break;
    case 2: __asm__("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD" : "=s"(m) : "v"(x), "v"(zero)); break;
    default: __asm__("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_3 src1_sel:DWORD" : "=s"(m) : "v"(x), "v"(zero)); break;
  }
  return m;
}
// Selects on such a mask: __builtin_amdgcn_inverse_ballot_w64 hands the SGPR pair to the compiler as a per-lane
// condition, so the select is the compiler's own pair of v_cndmask_b32 (zero / negated alternatives folded in) --
// and being instructions hipcc knows, it inserts the wait states an MFMA reading the result needs itself.  (Round 2
// wrote them as inline assembly: a select two instructions in front of an MFMA then went unprotected, found as
// run-to-run differences in the single-launch kernel.)
__device__ __forceinline__ double sel64(lanemask_t m, double a, double b) { return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b; }
__device__ __forceinline__ double sel64_or0(lanemask_t m, double a) { return __builtin_amdgcn_inverse_ballot_w64(m) ? a : 0.0; }
__device__ __forceinline__ double sel64_0or(lanemask_t m, double b) { return __builtin_amdgcn_inverse_ballot_w64(m) ? 0.0 : b; }
// m ? a : -b is the one select that stays inline assembly: the sign flip rides on the second
// v_cndmask_b32 as a source modifier, where hipcc spends a v_xor_b32 of its own (it splits the f64 select into 32-bit
// halves only after the negation has become an instruction; written on the halves by hand, with a float negation for
// the modifier to absorb, the select comes back as divergent BRANCHES) -- one vector instruction per entry of every W
// sweep.  Its result is an MFMA operand, so the two wait states are the caller's business again: the pass kernel has
// the product loop's LDS reads in between, the single-launch kernel an s_nop tied to the operand, and `make` checks
// every build (tools/check_asm_mfma_hazard.py) and removes a library that gets it wrong.
__device__ __forceinline__ double sel64_neg(lanemask_t m, double a, double b) {
  const uint32_t alo = (uint32_t)__double2loint(a), ahi = (uint32_t)__double2hiint(a);
  const uint32_t blo = (uint32_t)__double2loint(b), bhi = (uint32_t)__double2hiint(b);
  uint32_t lo, hi;
  __asm__("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(blo), "v"(alo), "s"(m));
  __asm__("v_cndmask_b32_e64 %0, -%1, %2, %3" : "=v"(hi) : "v"(bhi), "v"(ahi), "s"(m));   // the float negate modifier flips bit 31
  return mk_double(lo, hi);
}

__device__ __forceinline__ double wave_sum(double v) {
  // fixed butterfly order -> bitwise reproducible
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

#include "nbmf_pass_kernel.inc"
#include "nbmf_update_kernels.inc"
#include "nbmf_pack_kernels.inc"
#include "nbmf_peer_kernels.inc"
#include "nbmf_small_kernel.inc"

}  // namespace

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct nbmf_ctx {
  int device = 0;
  int cus = 256;            // compute units of the device (read once at creation)
  hipStream_t stream = nullptr;
  int64_t m = 0, n = 0, mA = 0, nA = 0;
  int k = 0, KP = 0, KB = 0;
  int KS = 1;               // slices of SLICE_K components (n_components > 128); KP = SLICE_K * KS then
  double* theta = nullptr;  // KS > 1: Theta in tile order, shared by the slices' sweeps
  int data_kind = -1;   // -1 = nothing uploaded
  int storage = NBMF_STORAGE_AUTO;   // storage path the next nbmf_upload is held to (nbmf_set_storage)
  void *dataA = nullptr, *dataB = nullptr, *maskA = nullptr, *maskB = nullptr;
  unsigned long long *bitsA = nullptr, *bitsB = nullptr;   // binary path: the lane-mask images the sweeps read (built from the codes)
  double n_obs = 0, n_obs_global = 0;
  double* rowcnt = nullptr;
  double *Wn = nullptr, *WT = nullptr, *WG = nullptr, *Hn = nullptr, *HT = nullptr, *HG = nullptr;
  bool have_factors = false;
  bool padB_ones = false;          // binary data observed everywhere: image B's pad rows are "ones" in its lane masks (mark_pad_rows_b)
  bool w_free = false;             // inside nbmf_w_only_steps (the W sweeps' variant, w_pass_args)
  bool factors_in_range = false;   // W >= 0 with column sums <= 1 + 1e-12, 0 <= H <= 1 - 1e-9 when they were set: Theta stays in [0, 1) for good
  int chunksH = 0, CH_H = 0, chunksW = 0, CH_W = 0;
  int *cstartH = nullptr, *cstartW = nullptr;   // device: chunk boundaries of the two sweeps
  double *slabH = nullptr, *slabW = nullptr, *Pbuf = nullptr, *lossbuf = nullptr, *prior = nullptr, *scal = nullptr;
  double* lossfin = nullptr;   // the slots of the fused loss assembly (PassFin): LL_EMPTY in every slot between sweeps
  int n_prior_blocks = 0;
  int* flags = nullptr;
  double* losses_d = nullptr;
  int losses_cap = 0;
  double* stage = nullptr;   // host->device staging for factors
  size_t stage_bytes = 0;
  unsigned long long* stats = nullptr;
  double alpha = 1.2, beta = 1.2, eps = 1e-8;
  int projection = NBMF_PROJ_NORMALIZE;
  // comm: RCCL communicator, or a host-mediated all-reduce callback (tests / fallback transport)
  void* comm = nullptr;
  nbmf_host_allreduce_fn host_reduce = nullptr;
  void* host_reduce_user = nullptr;
  double* host_buf = nullptr;   // pinned
  size_t host_buf_count = 0;
  int nranks = 1, rank = 0;
  // which internal axis is split over the ranks: 0 = rows of Y (W local, H replicated, exchange in the
  // H-step); 1 = columns of Y (H local, W replicated, exchange in the W-step)
  int shard_axis = 0;
  double n_div_global = 0;      // axis 1: global number of internal columns (the "/ n" of _solver.py:54)
  double* Qbuf = nullptr;       // axis 1: reduced W-step bracket [KP][mA], the all-reduce payload
  double* sbuf = nullptr;       // scalar exchange slot: [loglik, prior A, prior B]
  const double* ll_ptr = nullptr;   // where the most recent sweep left the (global) log-likelihood
  // axis 0: the K x N exchange is cut into column panels so that the second panel's all-reduce runs (on
  // stream2) while the first panel's H-update and its share of the W-pass compute
  int npanel = 1;
  int panels_wanted = 0;            // nbmf_set_exchange_panels: 0 = default / environment
  double peer_timeout_ms = 0;       // nbmf_set_peer_timeout_ms: 0 = default / environment
  long long pc0[3] = {0, 0, 0};     // panel column boundaries
  size_t pbase[2] = {0, 0};         // panel-major offsets into Pbuf: [P1_p (KP x wp) | P2_p (KP x wp)]
  size_t ll_index = 0;              // loglik slot, right behind panel 0 (travels with it)
  int wsplit = 0;                   // W-pass chunks [0, wsplit) read only panel 0's columns of H
  std::vector<int> bW_host;         // host copy of the W-pass chunk boundaries
  hipStream_t stream2 = nullptr;
  hipEvent_t evH = nullptr, evF = nullptr, ev1 = nullptr;
  // Beta log-prior partial sums the next loss assembly reads (the local array, or the arena's copy of the
  // most recent peer H-step)
  const double* prior_src = nullptr;
  int n_prior_src = 0;
  // peer transport (nbmf_peer_kernels.inc): own arena + flag block, every rank's mapping of them
  double* arena = nullptr;
  unsigned long long* pflags = nullptr;
  size_t arena_doubles = 0;
  int arena_axis = -1;
  struct ArenaSlotRef { void* p = nullptr; } arena_slot;   // pool entry behind arena / pflags (ArenaSlot*)
  bool peer = false;
  PeerView pv{};
  std::vector<void*> peer_mapped;   // hipIpcOpenMemHandle results to close
  unsigned long long epoch = 0, hseq = 0;
  long long offHX = 0, offPR = 0, offSC = 0, x_doubles = 0;
  long long sl_c0 = 0, sl_wp = 0;   // axis 0: the column slice of H this rank updates
  long long psl_c0[2] = {0, 0}, psl_wp[2] = {0, 0};   // ... per column panel when the exchange runs in two panels
  double *Pbuf_own = nullptr, *sbuf_own = nullptr, *Qbuf_own = nullptr;   // the private buffers while the arena stands in
  // the whole-fit-in-one-launch path for small problems (nbmf_small_kernel.inc): second parity of the factor
  // images, hand-off words, a snapshot of the factors for the fall-back
  struct SmallWs {
    double *Wn = nullptr, *WT = nullptr, *WG = nullptr, *Hn = nullptr, *HT = nullptr, *HG = nullptr;
    double *snapW = nullptr, *snapH = nullptr, *ll_part = nullptr, *prior_part = nullptr;
    unsigned long long* sync = nullptr;
    int* result = nullptr;
    double *part_buf = nullptr, *part_ll = nullptr;   // hand-off of split strips
    unsigned long long* part_flag = nullptr;
    char* slab = nullptr;        // the one allocation behind all of the above
    bool ready = false, disabled = false;
    int runs = 0, aborted = 0;   // statistics (nbmf_small_stats)
  } small;
  // batched runs (nbmf_run_batch): per-problem factor images / hand-off words for as many problems as fit the chip at once
  struct SmallBatchWs {
    char* slab = nullptr;          // [cap] per-problem blocks, `stride` bytes apart (small_batch_layout)
    double* table = nullptr;       // [2][cap] alpha - 1 | beta - 1 of the problems of a launch
    double* io = nullptr;          // staging for the factors of a whole batch (up and down)
    int cap = 0, losses_cap = 0;
    size_t io_doubles = 0;
    int launches = 0, problems = 0;   // statistics
  } small_batch;
  // progress reports out of nbmf_run (nbmf_set_progress)
  nbmf_progress_fn progress = nullptr;
  void* progress_user = nullptr;
  int progress_every = 0;
  std::atomic<int> cancelled{0};   // nbmf_cancel (any thread): sticky; runs return NBMF_ERR_STATE at the next iteration boundary
  // timing
  bool timing = false;
  int timing_stride = 1;   // events on the sweeps of every timing_stride-th iteration of a run (nbmf_timing_enable)
  int timing_it = 0;       // the iteration being enqueued
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<int> ev_kind;     // 0 = H-pass, 1 = W-pass
  size_t ev_used = 0;
  double t_ms[2] = {0, 0};
  int t_n[2] = {0, 0};
};

namespace {

// ---- device memory: a small caching allocator ---------------------------------------------------
// A fit of one of the reference's datasets is over in 2-3 ms; creating and destroying its context cost more than
// that in hipMalloc / hipFree calls (measured: 1.9 + 1.4 ms at 100 x 500, K = 6).  Blocks of up to 64 MiB are
// therefore kept on release -- per device, by exact size, up to NBMF_POOL_MB (default 1024) in total -- and handed to
// the next context that asks for that size.  A context synchronises its stream before it releases anything, so a
// reused block is never still in use; nothing relies on fresh memory being zero (every buffer is written or cleared
// before it is read).
struct DevPool {
  std::mutex mu;
  struct Key {
    int device;
    size_t bytes;
    bool operator<(const Key& o) const { return device != o.device ? device < o.device : bytes < o.bytes; }
  };
  std::multimap<Key, void*> free_blocks;
  std::map<void*, Key> live;     // every block handed out (pooled or not), for dfree to find its size
  size_t held = 0;
} g_pool;

template <typename T>
hipError_t dmalloc(T** out, size_t bytes) {
  *out = nullptr;
  if (bytes == 0) bytes = 256;
  int dev = 0;
  hipGetDevice(&dev);
  const DevPool::Key key{dev, (bytes + 255) / 256 * 256};
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.free_blocks.find(key);
    if (it != g_pool.free_blocks.end()) {
      *out = (T*)it->second;
      g_pool.held -= key.bytes;
      g_pool.free_blocks.erase(it);
      g_pool.live[(void*)*out] = key;
      return hipSuccess;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, key.bytes);
  if (e != hipSuccess) {   // out of memory: give back what is held and try once more
    (void)hipGetLastError();   // (the failed attempt's code must not outlive a retry that succeeds: see fail())
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> lk(g_pool.mu);
      for (auto& kv : g_pool.free_blocks) drop.push_back(kv.second);
      g_pool.free_blocks.clear();
      g_pool.held = 0;
    }
    for (void* q : drop) hipFree(q);
    e = hipMalloc(&p, key.bytes);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lk(g_pool.mu);
  g_pool.live[p] = key;
  *out = (T*)p;
  return hipSuccess;
}

hipError_t dfree(void* p) {
  if (!p) return hipSuccess;
  static const size_t cap = [] {
    const char* e = getenv("NBMF_POOL_MB");
    return (size_t)(e ? std::max(0, atoi(e)) : 1024) << 20;
  }();
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.live.find(p);
    if (it != g_pool.live.end()) {
      const DevPool::Key key = it->second;
      g_pool.live.erase(it);
      if (key.bytes <= (64u << 20) && g_pool.held + key.bytes <= cap) {
        g_pool.free_blocks.emplace(key, p);
        g_pool.held += key.bytes;
        return hipSuccess;
      }
    }
  }
  return hipFree(p);
}

// ... and the same for what else a context needs from the runtime: device properties are read once per device and
// process (hipGetDeviceProperties takes ~1 ms), streams are kept on release and reused (create + destroy ~1 ms).
struct DevInfo {
  bool known = false, gfx950 = false;
  int cus = 256;
  char arch[64] = {0};
};
DevInfo g_devinfo[64];
std::mutex g_devinfo_mu;

hipError_t device_info(int device, DevInfo* out) {
  std::lock_guard<std::mutex> lk(g_devinfo_mu);
  DevInfo& d = g_devinfo[device & 63];
  if (!d.known) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return e;
    d.known = true;
    d.gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    d.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    strncpy(d.arch, prop.gcnArchName, sizeof d.arch - 1);
  }
  *out = d;
  return hipSuccess;
}

std::vector<hipStream_t> g_streams[64];
std::mutex g_streams_mu;

hipError_t stream_acquire(int device, hipStream_t* out) {
  {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    auto& v = g_streams[device & 63];
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void stream_release(int device, hipStream_t st) {   // the caller has synchronised it
  std::lock_guard<std::mutex> lk(g_streams_mu);
  auto& v = g_streams[device & 63];
  if (v.size() < 64)
    v.push_back(st);
  else
    hipStreamDestroy(st);
}

// Wait for a stream whose work lasts milliseconds.  hipStreamSynchronize spins briefly and then sleeps on the
// completion interrupt; about one wake-up in 70 arrives ~8.6 ms late (tools/bimodal_probe.py: the kernel's own clock
// shows the same 35.0 ms on the device for a 3000-iteration fit whether the host saw 35.2 or 43.7 ms -- what round 2
// took for a second, slower "speed" of the single-launch path).  For a fit that lasts 3 ms that is the run three times
// over, so the small-problem paths poll the stream for up to `spin_ms` before they fall back to the blocking call.
hipError_t stream_wait_spin(hipStream_t st, double spin_ms = -1.0) {
  // (the budget: 200 ms unless NBMF_SPIN_MS says otherwise -- a host whose cores are shared by many rank threads may
  //  prefer to sleep at once: NBMF_SPIN_MS=0)
  static const double spin_default = [] {
    const char* e = getenv("NBMF_SPIN_MS");
    return e ? std::max(0.0, atof(e)) : 200.0;
  }();
  if (spin_ms < 0.0) spin_ms = spin_default;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > spin_ms) break;
  }
  return hipStreamSynchronize(st);
}

// ---- RCCL, loaded lazily so that single-GPU use has no dependency on it ---------------------
struct Uid {
  char internal[128];
};
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /*ncclUniqueId by value*/ Uid, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;      // optional: how many ranks the communicator itself says it joins (nbmf_comm_info)
  int (*CommCuDevice)(void*, int*) = nullptr;   // optional: the device it is bound to
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return NBMF_OK;
  // Which librccl: the one that belongs to the HIP runtime THIS library is linked against (the file next to
  // that libamdhip64), not whatever a plain soname lookup finds first -- a PyTorch wheel in the same process
  // brings its own, older copy bound to its own runtime.  NBMF_RCCL_LIBRARY overrides; the soname is the last
  // resort.
  std::vector<std::string> names;
  if (const char* e = getenv("NBMF_RCCL_LIBRARY")) names.push_back(e);
  Dl_info info;
  if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash + 1);
      names.push_back(dir + "librccl.so.1");
      names.push_back(dir + "librccl.so");
    }
  }
  for (const char* nm : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"}) names.push_back(nm);
  void* lib = nullptr;
  std::string tried;
  for (const std::string& nm : names) {
    lib = dlopen(nm.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (lib) {
      if (getenv("NBMF_DEBUG")) fprintf(stderr, "[nbmf] RCCL: %s\n", nm.c_str());
      break;
    }
    tried += nm + " ";
  }
  if (!lib) return fail(NBMF_ERR_COMM, "cannot load librccl (tried %s): %s", tried.c_str(), dlerror());
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, Uid, int))dlsym(lib, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
  g_rccl.CommCount = (int (*)(void*, int*))dlsym(lib, "ncclCommCount");
  g_rccl.CommCuDevice = (int (*)(void*, int*))dlsym(lib, "ncclCommCuDevice");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
    return fail(NBMF_ERR_COMM, "librccl is missing expected symbols");
  g_rccl.lib = lib;
  return NBMF_OK;
}
constexpr int kNcclFloat64 = 8;   // ncclDouble (rccl.h ncclDataType_t)
constexpr int kNcclSum = 0;       // ncclSum

#define NCCLCHK(call)                                                                                     \
  do {                                                                                                    \
    int r_ = (call);                                                                                      \
    if (r_ != 0)                                                                                          \
      return fail(NBMF_ERR_COMM, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
  } while (0)

// ---- peer-transport arenas: a process-wide pool ---------------------------------------------------
// Memory that has been exported through HIP IPC is never given back to the allocator while the process lives:
// a context returns its arena (payload block + flag block + their IPC handles) to this pool and the next
// context of a fitting size takes it over.  Freeing an exported arena and exporting a fresh allocation that
// lands on the same address handed the peers a mapping of the OLD memory (seen as a wrong factor in the second
// of two sharded fits in one process); a handle that stays valid for good cannot go stale.
struct PeerOrigin {   // tail of a handle block
  unsigned long long pid, arena, flags;   // pid: not the process id (two containers of a node may both run a pid 1) but a
  int device, reserved;                   // 64-bit number drawn once per process: process_nonce()
};
// Who exported a handle block: a rank of THIS process is addressed through the exporter's own pointers, so "this
// process" must not be decided by something two processes can share.  Process ids are unique per PID namespace only;
// a random 64-bit number drawn once at first use (mixed with the pid and the clock in case the random device is a
// constant) is unique for practical purposes.
unsigned long long process_nonce() {
  static const unsigned long long nonce = [] {
    unsigned long long v = 0;
    if (FILE* f = fopen("/dev/urandom", "rb")) {
      if (fread(&v, sizeof v, 1, f) != 1) v = 0;
      fclose(f);
    }
    v ^= (unsigned long long)getpid() * 0x9E3779B97F4A7C15ull;
    v ^= (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 0xBF58476D1CE4E5B9ull;
    return v ? v : 1ull;
  }();
  return nonce;
}
struct ArenaSlot {
  int device;
  size_t doubles;
  double* arena;
  unsigned long long* flags;
  char handles[NBMF_PEER_HANDLE_BYTES];
  bool in_use;
};
std::vector<ArenaSlot*> g_arenas;
std::mutex g_arena_mu;

int arena_acquire(int device, size_t doubles, ArenaSlot** out) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  ArenaSlot* best = nullptr;
  for (ArenaSlot* a : g_arenas)
    if (!a->in_use && a->device == device && a->doubles >= doubles && a->doubles <= 4 * doubles + (1u << 20) &&
        (!best || a->doubles < best->doubles))
      best = a;
  if (!best) {
    ArenaSlot* a = new ArenaSlot();
    a->device = device;
    a->doubles = doubles;
    a->arena = nullptr;
    a->flags = nullptr;
    a->in_use = false;
    hipError_t e = hipExtMallocWithFlags((void**)&a->arena, doubles * sizeof(double), hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&a->flags, PF_WORDS * sizeof(unsigned long long), hipDeviceMallocUncached);
    hipIpcMemHandle_t h[2];
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h[0], a->arena);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h[1], a->flags);
    if (e != hipSuccess) {
      if (a->arena) hipFree(a->arena);   // never exported: safe to free
      if (a->flags) hipFree(a->flags);
      delete a;
      return fail(NBMF_ERR_HIP, "peer arena: %s (allocation or hipIpcGetMemHandle)", hipGetErrorString(e));
    }
    // handle block: two IPC handles, then who exported them and where the memory sits in that process (a rank of
    // the same process cannot open its own process's handles and uses the addresses instead)
    static_assert(2 * sizeof(hipIpcMemHandle_t) + sizeof(PeerOrigin) <= NBMF_PEER_HANDLE_BYTES, "handle block too small");
    memset(a->handles, 0, sizeof a->handles);
    memcpy(a->handles, h, sizeof h);
    const PeerOrigin origin{process_nonce(), (unsigned long long)a->arena, (unsigned long long)a->flags, device, 0};
    memcpy(a->handles + sizeof h, &origin, sizeof origin);
    g_arenas.push_back(a);
    best = a;
  }
  best->in_use = true;
  *out = best;
  return NBMF_OK;
}

void arena_release(ArenaSlot* a) {
  if (!a) return;
  std::lock_guard<std::mutex> lk(g_arena_mu);
  a->in_use = false;
  if (const char* e = getenv("NBMF_ARENA_POOL"))
    if (atoi(e) == 0) {   // diagnostic: the old behaviour (free on release), to reproduce the stale-mapping failure
      hipFree(a->arena);
      hipFree(a->flags);
      g_arenas.erase(std::find(g_arenas.begin(), g_arenas.end(), a));
      delete a;
    }
}

// ---- the general path's logarithm table: one copy per device and process ------------------------
// Built when the first context of a device is created (nbmf_create, on that context's stream, which it then waits for --
// creation synchronises anyway), never in the launch path: a first-time build there would allocate and wait for the whole
// device under a process-wide lock -- illegal inside a stream capture, and a stall when another rank's kernel on the same
// device is itself waiting, in-kernel, for this rank (peer transport).  Launches only read the pointer.
std::atomic<const double2*> g_logtab[64];
std::mutex g_logtab_mu;
hipError_t log_table_build(int dev, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_logtab_mu);
  if (g_logtab[dev & 63].load(std::memory_order_acquire)) return hipSuccess;
  double2* t = nullptr;
  hipError_t e = hipMalloc((void**)&t, 1024 * sizeof(double2));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(log_table_kernel, dim3(4), dim3(256), 0, st, t);
  if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) {
    hipFree(t);
    return e;
  }
  g_logtab[dev & 63].store(t, std::memory_order_release);
  return hipSuccess;
}
// (the launch path: a read; hipErrorNotInitialized if no context of this device was ever created)
hipError_t log_table_device(const double2** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  *out = g_logtab[dev & 63].load(std::memory_order_acquire);
  return *out ? hipSuccess : hipErrorNotInitialized;
}

// ---- pass launch ----------------------------------------------------------------------------
// Timing (nbmf_timing_enable): a sweep that is ONE launch carries its two events in its own dispatch packet
// (hipExtLaunchKernelGGL: start and end of that kernel, no barrier packets in front of and behind it, which at
// configs[1] cost ~4.5 us each, 5 % of the iteration); EvScope puts them here, the next pass launch of this thread
// takes them.  Sweeps of several launches are bracketed by recorded events as before.
thread_local hipEvent_t tl_attach_start = nullptr, tl_attach_stop = nullptr;
std::atomic<long long> g_full_w_launches{0}, g_ragged_launches{0}, g_loss_assembly_recoveries{0};   // nbmf_variant_stats
template <int KB, int DATA, int MODE, int TH, bool TINY, bool RAG = false, bool FULL = false>
hipError_t launch_pass_tt(const PassArgs& a_, int chunks, hipStream_t st) {
  if (FULL) g_full_w_launches.fetch_add(1, std::memory_order_relaxed);
  if (RAG) g_ragged_launches.fetch_add(1, std::memory_order_relaxed);
  dim3 grid(a_.Cb / WG_WAVES, chunks);
  constexpr int lds_bytes = pass_lds_bytes(KB, DATA, MODE);
  if (lds_bytes > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)pass_kernel<KB, DATA, MODE, TH, TINY, RAG, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
  }
  PassArgs a = a_;
  if (DATA != DATA_BIN && MODE != MODE_W && MODE != MODE_T && pass_log_bits(KB) == 10) {
    hipError_t e = log_table_device(&a.ltab_g);
    if (e != hipSuccess) return e;
  }
  // (NBMF_PASS_TRACE=<n>: every n-th launch of each sweep kernel is traced -- with n > 1 the launches in between run back to
  //  back, so the traced one meets the chip in the state a real run leaves it in: clocks under sustained load)
  static const int trace_every = getenv("NBMF_PASS_TRACE") ? std::max(1, atoi(getenv("NBMF_PASS_TRACE"))) : 0;
  static std::atomic<long> trace_count{0};   // (per instantiation)
  if (trace_every && MODE != MODE_T && (trace_count.fetch_add(1) % trace_every) == trace_every - 1) {
    // diagnosis: where the workgroups of THIS launch spend their time outside the loop.  Every workgroup notes the wall
    // clock (100 MHz) at entry, at the top of its loop, at the loop's end and at exit; printed: the launch's span, the
    // spread of the entries and exits, mean prologue / loop / epilogue.  Serialises the stream (one launch at a time).
    const size_t n_wg = (size_t)grid.x * grid.y;
    unsigned long long* tr = nullptr;
    hipError_t e = dmalloc(&tr, sizeof(unsigned long long) * 8 * n_wg);
    if (e != hipSuccess) return e;
    a.trace = tr;
    hipLaunchKernelGGL((pass_kernel<KB, DATA, MODE, TH, TINY, RAG, FULL>), grid, dim3(64 * WG_WAVES), lds_bytes, st, a);
    std::vector<unsigned long long> h(8 * n_wg);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(h.data(), tr, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    dfree(tr);
    if (e != hipSuccess) return e;
    unsigned long long t_first = ~0ull, t_last = 0, last_entry = 0, first_exit = ~0ull;
    double pro = 0, loop = 0, epi = 0, epi_loss = 0, epi_slab = 0, loop_clk = 0;
    std::vector<double> exits;
    size_t live = 0;
    for (size_t i = 0; i < n_wg; ++i) {
      const unsigned long long* t = h.data() + 8 * i;
      if (!t[3]) continue;
      ++live;
      t_first = std::min(t_first, t[0]);
      t_last = std::max(t_last, t[3]);
      last_entry = std::max(last_entry, t[0]);
      first_exit = std::min(first_exit, t[3]);
      pro += (double)(t[1] - t[0]);
      loop += (double)(t[2] - t[1]);
      loop_clk += (double)t[7];
      epi += (double)(t[3] - t[2]);
      epi_loss += (double)(t[5] - t[2]);
      epi_slab += (double)(t[6] - t[5]);
      exits.push_back((double)t[3]);
    }
    if (live) {
      // mean time from entry to exit by the wave slot (HW_ID bits 3:0) the workgroup's first wave was given: does the
      // SIMD's arbiter favour some slots (oldest first)?
      double dur[16] = {0};
      int cnt[16] = {0};
      for (size_t i = 0; i < n_wg; ++i) {
        const unsigned long long* t = h.data() + 8 * i;
        if (!t[3]) continue;
        dur[t[4] & 15] += (double)(t[3] - t[0]);
        ++cnt[t[4] & 15];
      }
      fprintf(stderr, "[nbmf]   mean entry-to-exit by wave slot:");
      for (int w = 0; w < 16; ++w)
        if (cnt[w]) fprintf(stderr, " slot %d: %.1f us (%d)", w, dur[w] / cnt[w] * 0.01, cnt[w]);
      fprintf(stderr, "\n");
      std::sort(exits.begin(), exits.end());
      const double us = 0.01;   // 100 MHz ticks
      fprintf(stderr, "[nbmf] pass<K=%d,data=%d,mode=%d> %zu workgroups: span %.1f us | entries spread over %.1f us | exits: first %.1f, median %.1f, "
                      "last %.1f us before the end | per workgroup: prologue %.2f, loop %.2f, epilogue %.2f us (loss block %.2f, slab stores %.2f) | shader clock in the loops %.3f GHz\n", 16 * KB, DATA, MODE, live,
              (double)(t_last - t_first) * us, (double)(last_entry - t_first) * us, (double)(t_last - first_exit) * us,
              ((double)t_last - exits[exits.size() / 2]) * us, 0.0, pro / live * us, loop / live * us, epi / live * us, epi_loss / live * us,
              epi_slab / live * us, loop > 0 ? loop_clk / loop * 0.1 : 0.0);
    }
    return hipSuccess;
  }
  if (tl_attach_start) {
    hipEvent_t e0 = tl_attach_start, e1 = tl_attach_stop;
    tl_attach_start = tl_attach_stop = nullptr;
    hipExtLaunchKernelGGL((pass_kernel<KB, DATA, MODE, TH, TINY, RAG, FULL>), grid, dim3(64 * WG_WAVES), lds_bytes, st, e0, e1, 0, a);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((pass_kernel<KB, DATA, MODE, TH, TINY, RAG, FULL>), grid, dim3(64 * WG_WAVES), lds_bytes, st, a);
  return hipGetLastError();
}
// (eps below 1e-70 on the binary path: the variant with per-entry reciprocals and renormalisation, see pass_kernel)
template <int KB, int DATA, int MODE, int TH = 0>
hipError_t launch_pass_t(const PassArgs& a, int chunks, hipStream_t st) {
  if (MODE != MODE_T && a.tiny_eps) return launch_pass_tt<KB, DATA, MODE, TH, MODE != MODE_T>(a, chunks, st);
  // Fewer components than the layout holds -- 10 in the K = 16 layout, 40 in the K = 64 one: the variant that skips the
  // k-steps and 16-blocks of components that are padding (RAG, nbmf_pass_kernel.inc).  Where it pays (measured, 32768 x 8192,
  // byte codes: k = 4 +19 %, 8 +11-16 %, 10-12 +2-6 %, 20 +9 %, 24 +5 %, 40 +20 %, 48 +15 %): up to the K = 64 layout -- at
  // K = 128 the branches cost the register-tuned schedule more than the skipped MFMAs give back (k = 100: -4.6 %) --, when
  // at least a twelfth of the tile's MFMAs goes (k = 50: three of 48, -1.7 %), and not on the general path at K <= 16 (four
  // tiles per stage with branches between their MFMAs: 250 registers and scratch, -4 % at k = 10).
  constexpr bool HAS_RAG = TH == 0 && MODE != MODE_T && KB <= 4 && (DATA == DATA_BIN || KB >= 2);
  // W sweeps over a matrix that is observed everywhere and has no pad rows in the swept dimension: the two-state variant
  // (FULL, nbmf_pass_kernel.inc; NBMF_NO_FULL_W=1: the three-state kernels on such data -- tests)
  constexpr bool HAS_FULL = TH == 0 && MODE == MODE_W && DATA == DATA_BIN;
  static const bool no_full = getenv("NBMF_NO_FULL_W") != nullptr;
  const bool full = HAS_FULL && a.full && !no_full;
  static const bool no_rag = getenv("NBMF_NO_RAGGED_K") != nullptr;   // (tests: the full-K kernels on a ragged K -- the same bits)
  if (HAS_RAG && !no_rag && a.ksteps > 0 && a.ksteps < 4 * KB) {
    const int per_block = MODE == MODE_H ? 8 : (MODE == MODE_W ? 4 : 0);   // back-product MFMAs of one 16-block of components
    const int skipped = (4 * KB - a.ksteps) + (KB >= 4 ? (KB - a.kblocks) * per_block : 0);
    const int total = 4 * KB + KB * per_block;
    if (12 * skipped >= total)
      return full ? launch_pass_tt<KB, DATA, MODE, TH, false, HAS_RAG, HAS_RAG && HAS_FULL>(a, chunks, st)
                  : launch_pass_tt<KB, DATA, MODE, TH, false, HAS_RAG>(a, chunks, st);
  }
  return full ? launch_pass_tt<KB, DATA, MODE, TH, false, false, HAS_FULL>(a, chunks, st)
              : launch_pass_tt<KB, DATA, MODE, TH, false>(a, chunks, st);
}

// n_components > 128: the sweep of one slice with Theta read from memory (TH = 1: whole; TH = 2: the earlier
// slices' part, completed and stored here), and the Theta sweep itself
template <int MODE, int TH>
hipError_t launch_pass_slice(int data_kind, const PassArgs& a, int chunks, hipStream_t st) {
  switch (data_kind) {
    case DATA_BIN: return launch_pass_t<8, DATA_BIN, MODE, TH>(a, chunks, st);
    case DATA_F64: return launch_pass_t<8, DATA_F64, MODE, TH>(a, chunks, st);
    case DATA_F64M: return launch_pass_t<8, DATA_F64M, MODE, TH>(a, chunks, st);
  }
  return hipErrorInvalidValue;
}
inline hipError_t launch_theta(const PassArgs& a, int chunks, hipStream_t st) {
  return launch_pass_t<8, DATA_BIN, MODE_T, 0>(a, chunks, st);
}

template <int DATA, int MODE>
hipError_t launch_pass_kb(int KB, const PassArgs& a, int chunks, hipStream_t st) {
  switch (KB) {
    case 1: return launch_pass_t<1, DATA, MODE>(a, chunks, st);
    case 2: return launch_pass_t<2, DATA, MODE>(a, chunks, st);
    case 4: return launch_pass_t<4, DATA, MODE>(a, chunks, st);
    case 8: return launch_pass_t<8, DATA, MODE>(a, chunks, st);
  }
  return hipErrorInvalidValue;
}

template <int MODE>
hipError_t launch_pass(int KB, int data_kind, const PassArgs& a, int chunks, hipStream_t st) {
  switch (data_kind) {
    case DATA_BIN: return launch_pass_kb<DATA_BIN, MODE>(KB, a, chunks, st);
    case DATA_F64: return launch_pass_kb<DATA_F64, MODE>(KB, a, chunks, st);
    case DATA_F64M: return launch_pass_kb<DATA_F64M, MODE>(KB, a, chunks, st);
  }
  return hipErrorInvalidValue;
}

template <int DATA, int MODE>
const void* pass_ptr_kb(int KB) {
  switch (KB) {
    case 1: return (const void*)pass_kernel<1, DATA, MODE>;
    case 2: return (const void*)pass_kernel<2, DATA, MODE>;
    case 4: return (const void*)pass_kernel<4, DATA, MODE>;
    case 8: return (const void*)pass_kernel<8, DATA, MODE>;
  }
  return nullptr;
}

template <int MODE>
const void* pass_ptr(int KB, int data_kind) {
  switch (data_kind) {
    case DATA_BIN: return pass_ptr_kb<DATA_BIN, MODE>(KB);
    case DATA_F64: return pass_ptr_kb<DATA_F64, MODE>(KB);
    case DATA_F64M: return pass_ptr_kb<DATA_F64M, MODE>(KB);
  }
  return nullptr;
}

// workgroups of this pass kernel that one CU holds at once (registers and LDS decide)
template <int MODE>
int resident_per_cu(int KB, int data_kind) {
  int n = 0;
  const void* f = pass_ptr<MODE>(KB, data_kind);
  const int lds_bytes = pass_lds_bytes(KB, data_kind, MODE);
  if (!f || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 64 * WG_WAVES, lds_bytes) != hipSuccess || n < 1) n = 2;
  return std::min(n, 8);
}

// Split a sweep of Rb row blocks into chunks: aim for ~2048 workgroups (several rounds over the 512-768
// resident ones, so the dispatcher can even out slow workgroups) without making the sweeps shorter than
// 64 blocks (each workgroup pays a fixed prologue/epilogue and writes its own partial slab).
// Measured at c3 / K=64: 768 workgroups = exactly one round of the 768 resident slots is SLOWER (3.23 ms)
// than 2048 (3.15 ms); anything from 1536 to 4608 is within 1 %.  `slots` is kept for diagnostics.
// NBMF_TARGET_WGS=<n> overrides the target (tuning experiments only).
void pick_chunks(int strips_groups, int Rb, int NB, int slots, int ns, int cus, bool one_round_ok, int* chunks, int* CH) {
  (void)slots;
  int target = 2048;
  const char* te = getenv(one_round_ok ? "NBMF_TARGET_WGS_H" : "NBMF_TARGET_WGS_W");   // (tuning experiments: one sweep only)
  if (!te) te = getenv("NBMF_TARGET_WGS");
  if (te) target = std::max(1, atoi(te));
  int want = (target + strips_groups - 1) / strips_groups;
  const int max_chunks = std::max(1, Rb / NB);
  want = std::min(std::max(want, 1), max_chunks);
  int ch = (Rb + want - 1) / want;
  // (small problems -- fewer than 512 workgroups even at 64-block chunks -- are bound by the length of the
  //  sweep a single workgroup walks, not by per-workgroup overhead: 8-block chunks there.  BASELINE configs[0],
  //  100 x 500 with K = 6: 37.4 -> 26.2 us per iteration (26 700 -> 38 100 iterations/s); the 36-fit
  //  perplexity grid on the lastfm-sized matrix 0.63 -> 0.33 s.)
  // (ns = strips per wave: a workgroup of the two-strip kernels does a 64-block sweep's work in 32 blocks)
  int min_blocks = (long long)strips_groups * ns * ((Rb + 63) / 64) >= 512 ? 64 / ns : 8;
  if (const char* e = getenv("NBMF_MIN_BLOCKS")) min_blocks = std::max(NB, atoi(e));   // (tuning experiments only)
  const bool clipped = ch < std::min(Rb, min_blocks);   // the sweep is too short for the target at min_blocks per chunk
  ch = std::max(ch, std::min(Rb, min_blocks));
  ch = (int)round_up(ch, NB);
  // The last round.  Workgroups of a sweep are equally long, so they finish in rounds; a CU is as fast with two
  // resident workgroups as with three (the MFMA pipe is full either way), but a last round that leaves CUs EMPTY is
  // lost time: 266 strip groups x 8 chunks = 2128 workgroups on 512 such places (configs[4]'s W-pass, K = 128) are
  // 4.16 rounds -- a fifth round for 80 workgroups, MFMA-busy 84 %.  More chunks are tried (up to four times as many,
  // whole multiples of 8 first: the XCD renumbering of pass_kernel wants those) until the rounds are >= 95 % full.
  // The sweeps of configs[1..3] and of their shards already are (2048 or 1024 workgroups: 4 or 2 full rounds).
  if (cus > 0 && !te && !getenv("NBMF_MIN_BLOCKS") && !getenv("NBMF_NO_ROUND_FILL")) {
    const long long places = (long long)cus * std::min(std::max(slots / cus, 1), 2);
    auto fill = [&](int blocks_per_chunk) {   // how full the rounds are with chunks of that many row blocks
      const long long wgs = (long long)strips_groups * ((Rb + blocks_per_chunk - 1) / blocks_per_chunk);
      return (double)wgs / (double)(((wgs + places - 1) / places) * places);
    };
    int base = (Rb + ch - 1) / ch;   // chunks so far
    // (short sweeps at K >= 64, where two workgroups per CU fill the MFMA pipe: ONE round of exactly that many beats two
    //  rounds of half-length workgroups -- the 8192-row shard of configs[2], 128 strip groups x 512 row blocks: 4 chunks
    //  1 443 it/s, 8 chunks 1 422; at K = 32, which wants four waves per SIMD, and for the long sweeps of configs[2]
    //  itself it is the other way round, by 1 %.  The H sweep only: a row-sharded run may launch its W sweep in two parts,
    //  one per exchange panel, and half of four chunks would leave one workgroup per CU)
    if (one_round_ok && clipped && NB <= 2 && places % strips_groups == 0 && places / strips_groups < base &&
        (Rb + (int)(places / strips_groups) - 1) / (int)(places / strips_groups) >= min_blocks) {
      const int n1 = (int)(places / strips_groups);
      ch = (int)round_up((Rb + n1 - 1) / n1, NB);
      base = (Rb + ch - 1) / ch;
    }
    if ((long long)strips_groups * base > places && fill(ch) < 0.95) {
      int best = ch;
      double best_fill = fill(ch);
      for (int pass = 0; pass < 2 && best_fill < 0.95; ++pass) {   // pass 0: whole multiples of 8 chunks only
        for (int n = base + 1; n <= 4 * base; ++n) {
          if (pass == 0 && (n & 7)) continue;
          const int cand = (int)round_up((Rb + n - 1) / n, NB);
          if (cand < std::min(Rb, min_blocks)) break;
          const double f = fill(cand);
          if (f > best_fill + 1e-9) {
            best_fill = f;
            best = cand;
            if (f >= 0.95) break;
          }
        }
      }
      ch = best;
    }
  }
  *CH = ch;
  *chunks = (Rb + ch - 1) / ch;
}

// Chunk boundaries (row blocks) of a sweep.  Equal chunks: cutting the tail of the sweep four times finer
// (so that the last, lonely workgroups on a CU are short) was measured and bought nothing (c3 H-pass 3.15
// vs 3.17 ms) while adding slabs for the update kernels to sum.
std::vector<int> chunk_boundaries(int Rb, int ch) {
  std::vector<int> b;
  b.push_back(0);
  for (int pos = 0; pos < Rb;) {
    pos = std::min(pos + ch, Rb);
    b.push_back(pos);
  }
  return b;
}

struct EvScope {
  nbmf_ctx* c;
  int kind;
  size_t slot = (size_t)-1;
  bool attach;
  // count = false: add the time to `kind` but do not count a launch (second part of a split sweep)
  // attach = true: the scope holds exactly one pass launch, on c->stream, which carries the events itself (launch_pass_tt)
  EvScope(nbmf_ctx* c_, int kind_, bool count = true, bool attach_ = false)
      : c(c_), kind(kind_ | (count ? 0 : 0x100)), attach(attach_ && !getenv("NBMF_TIMING_BRACKET")) {
    if (!c->timing || (c->timing_stride > 1 && c->timing_it % c->timing_stride != 0)) return;
    if (c->ev_used + 2 > c->ev.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->ev.push_back(e);
      }
      c->ev_kind.push_back(kind);
    } else {
      c->ev_kind[c->ev_used / 2] = kind;
    }
    slot = c->ev_used;
    c->ev_used += 2;
    if (attach) {
      tl_attach_start = c->ev[slot];
      tl_attach_stop = c->ev[slot + 1];
    } else {
      hipEventRecord(c->ev[slot], c->stream);
    }
  }
  ~EvScope() {
    if (slot == (size_t)-1) return;
    if (attach) {
      if (tl_attach_start) {   // (no launch took them: an error path)
        tl_attach_start = tl_attach_stop = nullptr;
        c->ev_used = slot;
      }
    } else {
      hipEventRecord(c->ev[slot + 1], c->stream);
    }
  }
};

void timing_collect(nbmf_ctx* c) {
  // caller has synchronised the stream
  for (size_t s = 0; s + 1 < c->ev_used; s += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev[s], c->ev[s + 1]) == hipSuccess) {
      c->t_ms[c->ev_kind[s / 2] & 0xFF] += ms;
      if (!(c->ev_kind[s / 2] & 0x100)) c->t_n[c->ev_kind[s / 2] & 0xFF] += 1;
    }
  }
  c->ev_used = 0;
}

// In-place sum over ranks of `count` doubles at device pointer `p`, on the context's stream.
int all_reduce_inplace(nbmf_ctx* c, double* p, size_t count, hipStream_t st = nullptr) {
  if (c->comm) {
    NCCLCHK(g_rccl.AllReduce(p, p, count, kNcclFloat64, kNcclSum, c->comm, st ? st : c->stream));
  } else if (c->peer) {
    const long long off = p - c->arena;
    if (off < 0 || (size_t)off + count > c->arena_doubles) return fail(NBMF_ERR_STATE, "internal: peer exchange outside the arena");
    hipStream_t s = st ? st : c->stream;
    const unsigned long long e = ++c->epoch;
    const long long slice = ((long long)count + c->pv.nranks - 1) / c->pv.nranks;
    const unsigned grid = (unsigned)std::min<long long>(128, std::max<long long>(1, (slice + 1023) / 1024));
    // (fault injection for the tests of the failure path: NBMF_PEER_FAULT=1 makes the LAST rank skip its part of
    //  every generic exchange, as a rank whose links do not work would -- the others must time out, not hang)
    static const bool fault = getenv("NBMF_PEER_FAULT") && atoi(getenv("NBMF_PEER_FAULT")) != 0;
    if (!(fault && c->pv.nranks > 1 && c->pv.rank == c->pv.nranks - 1))
      hipLaunchKernelGGL(peer_reduce_kernel, dim3(grid), dim3(256), 0, s, c->pv, e, off, (long long)count, c->flags);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, s, c->pv, e, c->flags);
    HIPCHK(hipGetLastError());
  } else if (c->host_reduce) {
    if (count > c->host_buf_count) return fail(NBMF_ERR_STATE, "internal: host exchange buffer too small");
    HIPCHK(hipMemcpyAsync(c->host_buf, p, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->host_reduce(c->host_reduce_user, c->host_buf, (int64_t)count) != 0)
      return fail(NBMF_ERR_COMM, "host all-reduce callback failed");
    HIPCHK(hipMemcpyAsync(p, c->host_buf, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  return NBMF_OK;
}

inline bool is_sharded(const nbmf_ctx* c) { return c->comm || c->host_reduce || c->peer; }

// what a sweep reads of the data: the lane-mask image on the binary path, the double tiles otherwise
inline const void* sweep_image(const nbmf_ctx* c, int image) {
  if (c->data_kind == DATA_BIN) return image == 0 ? (const void*)c->bitsA : (const void*)c->bitsB;
  return image == 0 ? c->dataA : c->dataB;
}

// column strips one workgroup of the pass kernels covers (one per wave)
inline int wg_strips(const nbmf_ctx*) { return WG_WAVES; }

// Every sweep over image A takes in all entries of the padded mA x nA grid.  A pad entry has Theta == 0 and
// counts as "not an observed one" (binary path: code 0; general path: y = 0, weight 0), so it contributes
// exactly log(fl(fl(1-0)+eps)) = log(1+eps) to the likelihood; their total is removed before the loss is
// assembled (and before any all-reduce).  A strictly masked sweep skips unobserved entries, pads included --
// except on the general path without a mask, where everything counts as observed.
// Which variant the sweeps over image A (H-pass, evaluation sweep) take on the binary path: the plain one forms the
// ratio block of an H sweep from |Theta - z| (pass_kernel), which IS the reference's arithmetic while
// 0 <= Theta < 1 -- true throughout a fit that starts from factors in range (W on the simplex, H <= 1 - eps stay so) and
// eps >= 1e-12 (so that 1 - eps is below 1).  Anything else -- H_init above 1, negative entries, a tiny eps -- takes the
// TINY variant, whose selects follow `_solver.py:42-43` for any Theta.
// Which variant of the sweeps a run takes.  The plain variant (TINY = false) forms |Theta - z| for the denominators,
// shares one reciprocal among a lane's entries and renormalises the likelihood's product once per trip; all of that is
// the reference's arithmetic to rounding (one-step goldens <= 1e-13, curves <= 1e-10: tests/test_gpu_parity.py) -- since
// round 4 no longer bit for bit on the binary path either: with eps riding in Theta's accumulator the H-update forms
// P1 = P1' - P2' / (1 + 2 eps) from all-entry sums, so a column with no observed one gets a rounding residue (which the
// clip to [eps, 1 - eps] absorbs) where the reference has an exact 0 -- and ONLY under these invariants:
//   eps >= 1e-12            (1 - eps < 1 in double; sixteen factors >= eps stay above 1e-192; products of four or eight
//                            denominators cannot underflow),
//   W >= 0 with column sums <= 1 + 1e-12  and  0 <= H <= 1 - 1e-9   when the factors were set (factors_in_range, checked
//                            on the device by nbmf_set_factors): then 0 <= Theta < 1 for the whole fit -- the W-update
//                            renormalises or projects onto the simplex, the H-update clips to [eps, 1 - eps].
// Anything else -- H_init above 1 (the reference uses it as given, _solver.py:133), negative entries, transform's
// un-normalised start (_base.py:175: w_free), a tiny eps -- takes the TINY variant, which keeps the reference's own
// selects, one reciprocal per entry and a renormalisation per entry
// (tests: test_factors_out_of_the_fits_range_follow_the_reference).
//   eps < 2^-22            (binary path only: the high word of 1 + 2 eps is that of 1.0, so that z = 1 + 2 eps shares it;
//                            the general path forms (1 + 2 eps) - t1 by plain arithmetic and has no such limit).
int tiny_a(const nbmf_ctx* c) {
  return c->eps < 1e-70 || !(c->eps >= 1e-12 && c->factors_in_range) || (c->data_kind == DATA_BIN && !(c->eps < 0x1p-22));
}
// In what form the H sweep of the binary path's plain variant leaves its products: P1' = L r (ALL entries'
// reciprocals) and P2' = (1 + 2 eps) L R2.  The H-update kernels take 1 / (1 + 2 eps) and undo both maps on the summed
// K x N products -- P2 = P2' / (1 + 2 eps), P1 = P1' - P2 (linear: after the sum over chunks and over ranks) --; 0 says
// the products are P1, P2 themselves (general path, the TINY variant, sliced runs).
double pform_inv12(const nbmf_ctx* c) {
  if (c->data_kind != DATA_BIN || c->KS != 1 || tiny_a(c)) return 0.0;
  return 1.0 / ((1.0 + c->eps) + c->eps);
}

double ll_pad_of(const nbmf_ctx* c, int strict = 0) {
  // (strictly masked sweeps count observed entries only, and a pad is never one: nothing to remove.  Until round 4 the
  //  8-byte path with a folded mask was exempted here although its pads -- NaN then, -0.0 now -- contribute nothing either:
  //  held-out perplexities of real-valued data with a binary mask came out 2e-8 off, found by
  //  test_real_valued_data_with_a_binary_mask_every_sweep_variant)
  if (strict) return 0.0;
  const double n_pad = (double)c->mA * (double)c->nA - (double)c->m * (double)c->n;
  return n_pad * log(1.0 + c->eps);
}

// What travels after a sweep over image A (H-pass or Theta-only sweep), and where the global
// log-likelihood ends up (c->ll_ptr):
//   single GPU        nothing; finalize sums the per-wave partials itself
//   axis 0 (rows)     Theta-only sweep: the loglik scalar (the H-pass exchange is enqueue_iteration_rows)
//   axis 1 (columns)  [loglik, prior A, prior B]: the products stay local, the scalars do not
int enqueue_exchange_after_sweep(nbmf_ctx* c, const PassArgs& a, bool with_products, int strict) {
  const int n_loss = c->chunksH * (a.Cb / wg_strips(c));   // one log-likelihood partial per workgroup
  const double pad = ll_pad_of(c, strict);
  if (!is_sharded(c)) {
    c->ll_ptr = nullptr;
    return NBMF_OK;
  }
  if (c->shard_axis == 0 && with_products)
    return fail(NBMF_ERR_STATE, "internal: the row-split H-step exchange is driven by enqueue_iteration_rows");
  hipLaunchKernelGGL(ll_reduce_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->lossbuf, n_loss, pad, c->sbuf,
                     c->flags);
  HIPCHK(hipGetLastError());
  size_t cnt = 1;
  if (c->shard_axis == 1) {
    hipLaunchKernelGGL(prior_reduce_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->prior, c->n_prior_blocks,
                       c->sbuf + 1, c->flags);
    HIPCHK(hipGetLastError());
    cnt = 3;
  }
  if (int rc = all_reduce_inplace(c, c->sbuf, cnt)) return rc;
  c->ll_ptr = c->sbuf;
  return NBMF_OK;
}

// ---- n_components > 128: slices of SLICE_K components --------------------------------
// Theta of the current factors WITHOUT the last slice's part into c->theta, in the tile order of image A (image
// 0: H-pass and loglik sweeps) or image B (image 1: W-pass): one Theta-only sweep per slice, each adding onto the
// previous ones.
int enqueue_theta(nbmf_ctx* c, int image) {
  for (int sl = 0; sl < c->KS - 1; ++sl) {   // (the last slice's part is added by that slice's own fused sweep)
    const size_t offW = (size_t)sl * SLICE_K * c->mA, offH = (size_t)sl * SLICE_K * c->nA;
    PassArgs a{};
    a.LT = image == 0 ? c->WT + offW : c->HT + offH;
    a.LG = image == 0 ? c->WG + offW : c->HG + offH;   // not read by this mode
    a.RfT = image == 0 ? c->HT + offH : c->WT + offW;
    a.theta = c->theta;
    a.accum = sl > 0;
    a.done = c->flags;
    a.Rb = (int)((image == 0 ? c->mA : c->nA) / 16);
    a.Cb = (int)((image == 0 ? c->nA : c->mA) / 16);
    a.chunk_start = image == 0 ? c->cstartH : c->cstartW;
    a.C_alloc = image == 0 ? c->nA : c->mA;
    a.eps = c->eps;
    a.tiny_eps = c->eps < 1e-70;
    HIPCHK(launch_theta(a, image == 0 ? c->chunksH : c->chunksW, c->stream));
  }
  return NBMF_OK;
}

// The sweeps over image A of a sliced run: back-products (with_products) of every slice, or the log-likelihood
// only.  Order: Theta-only sweeps of slices 0 .. KS-2; the LAST slice's sweep completes Theta from what they
// stored, stores the total, reports the likelihood and forms its own products; then the other slices' sweeps
// read the total.  (Theta crosses HBM 24 KS - 16 bytes per entry instead of 24 KS - 8, in one sweep fewer.)
int enqueue_a_sweeps_sliced(nbmf_ctx* c, bool with_products, int strict, int clip) {
  if (int rc = enqueue_theta(c, 0)) return rc;
  const size_t per = (size_t)SLICE_K * c->nA;
  const int last = c->KS - 1;
  for (int step = 0; step < (with_products ? c->KS : 1); ++step) {
    const int sl = step == 0 ? last : step - 1;
    PassArgs a{};
    a.data = sweep_image(c, 0);
    a.mask = c->maskA;
    a.LT = c->WT + (size_t)sl * SLICE_K * c->mA;
    a.LG = c->WG + (size_t)sl * SLICE_K * c->mA;
    a.RfT = c->HT + (size_t)sl * SLICE_K * c->nA;
    a.out1 = c->slabH + (size_t)sl * c->chunksH * per;
    a.out2 = c->slabH + (size_t)(c->KS + sl) * c->chunksH * per;
    a.lossbuf = step == 0 ? c->lossbuf : nullptr;   // the likelihood is the same in every slice's sweep
    a.done = c->flags;
    a.Rb = (int)(c->mA / 16);
    a.Cb = (int)(c->nA / 16);
    a.chunk_start = c->cstartH;
    a.C_alloc = c->nA;
    a.eps = c->eps;
    a.tiny_eps = tiny_a(c);
    a.ksteps = (c->k + 3) / 4;
    a.kblocks = (c->k + 15) / 16;
    a.strict = strict;
    a.clip = clip;
    a.theta = c->theta;
    hipError_t e;
    if (with_products)
      e = step == 0 ? launch_pass_slice<MODE_H, 2>(c->data_kind, a, c->chunksH, c->stream)
                    : launch_pass_slice<MODE_H, 1>(c->data_kind, a, c->chunksH, c->stream);
    else
      e = launch_pass_slice<MODE_L, 2>(c->data_kind, a, c->chunksH, c->stream);
    HIPCHK(e);
  }
  return NBMF_OK;
}

// Single GPU, one slice: the sweep that scores iteration t assembles its loss itself (PassFin: no finalize launch).
// every slot of the fused loss assembly empty (PassFin): at set-up, and whenever a sweep may have left some filled
// (a run that was interrupted by an error)
__global__ void fill_u64_kernel(unsigned long long* p, size_t n, unsigned long long v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int fill_ll_empty(nbmf_ctx* c) {
  const size_t n = (size_t)c->chunksH * (c->nA / 16 / WG_WAVES);
  if (!c->lossfin || !n) return NBMF_OK;
  hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (unsigned long long*)c->lossfin, n, LL_EMPTY);
  HIPCHK(hipGetLastError());
  return NBMF_OK;
}
bool fin_fusable(const nbmf_ctx* c) { return !is_sharded(c) && c->KS == 1 && !getenv("NBMF_NO_FUSED_FINALIZE"); }
void fin_fill(nbmf_ctx* c, PassArgs& a, int t, double tol, int strict) {
  a.fin.on = 1;
  a.fin.wait_ticks = LL_WAIT_TICKS;
  if (const char* e = getenv("NBMF_PASSFIN_FAULT"))   // tests: one workgroup withholds its partial, the wait is short
    if (atoi(e) == 2) {
      a.fin.wait_ticks = 0;             // (2: nobody withholds anything, but the assembling workgroup does not wait at all -- it
    } else if (atoi(e) != 0) {          //  gives up while other workgroups of its sweep are still running, as under a tenant)
      a.fin.on = 2;
      a.fin.wait_ticks = 20000000ull;   // 0.2 s
    }
  a.lossbuf = c->lossfin;   // (the sweep's workgroups hand their partials in here; see PassFin)
  a.fin.t = t;
  a.fin.n_ll = c->chunksH * (int)(c->nA / 16 / wg_strips(c));
  a.fin.n_prior = c->n_prior_src;
  a.fin.ll_pad = ll_pad_of(c, strict);
  a.fin.am1 = c->alpha - 1.0;
  a.fin.bm1 = c->beta - 1.0;
  a.fin.n_obs = c->n_obs_global;
  a.fin.tol = tol;
  a.fin.prior = c->prior_src;
  a.fin.losses = c->losses_d;
  a.fin.scal = c->scal;
  a.fin.flags = c->flags;
}

// fin_t >= 0 (and fin_fusable): also the loss and stop test of iteration fin_t, in the sweep's last workgroup
int enqueue_h_pass(nbmf_ctx* c, int fin_t = -1, double tol = 0.0) {
  if (c->KS > 1) {
    {
      EvScope ev(c, 0);
      if (int rc = enqueue_a_sweeps_sliced(c, true, 0, 0)) return rc;
    }
    PassArgs a{};
    a.Cb = (int)(c->nA / 16);
    return enqueue_exchange_after_sweep(c, a, /*with_products=*/true, /*strict=*/0);
  }
  PassArgs a{};
  a.data = sweep_image(c, 0);
  a.mask = c->maskA;
  a.LT = c->WT;
  a.LG = c->WG;
  a.RfT = c->HT;
  a.out1 = c->slabH;
  a.out2 = c->slabH + (size_t)c->chunksH * c->KP * c->nA;
  a.lossbuf = c->lossbuf;
  a.done = c->flags;
  a.Rb = (int)(c->mA / 16);
  a.Cb = (int)(c->nA / 16);
  a.chunk_start = c->cstartH;
  a.C_alloc = c->nA;
  a.eps = c->eps;
  a.tiny_eps = tiny_a(c);
  a.ksteps = (c->k + 3) / 4;
  a.kblocks = (c->k + 15) / 16;
  if (fin_t >= 0) fin_fill(c, a, fin_t, tol, 0);
  {
    EvScope ev(c, 0, true, /*attach=*/true);
    HIPCHK(launch_pass<MODE_H>(c->KB, c->data_kind, a, c->chunksH, c->stream));
  }
  if (int rc = enqueue_exchange_after_sweep(c, a, /*with_products=*/true, /*strict=*/0)) return rc;
  return NBMF_OK;
}

// Theta-only sweep (no back-products): the log-likelihood of the current factors at a third of the
// H-pass's MFMA work; the per-wave partials land in lossbuf exactly as an H-pass leaves them.
int enqueue_loglik_pass(nbmf_ctx* c, int strict, int clip = 0, int fin_t = -1, double tol = 0.0) {
  if (c->KS > 1) {
    if (int rc = enqueue_a_sweeps_sliced(c, false, strict, clip)) return rc;
    PassArgs a{};
    a.Cb = (int)(c->nA / 16);
    return enqueue_exchange_after_sweep(c, a, /*with_products=*/false, strict);
  }
  PassArgs a{};
  a.data = sweep_image(c, 0);
  a.mask = c->maskA;
  a.LT = c->WT;
  a.LG = c->WG;
  a.RfT = c->HT;
  a.out1 = c->slabH;      // unused
  a.out2 = nullptr;
  a.lossbuf = c->lossbuf;
  a.done = c->flags;
  a.Rb = (int)(c->mA / 16);
  a.Cb = (int)(c->nA / 16);
  a.chunk_start = c->cstartH;
  a.C_alloc = c->nA;
  a.eps = c->eps;
  a.tiny_eps = tiny_a(c);
  a.ksteps = (c->k + 3) / 4;
  a.kblocks = (c->k + 15) / 16;
  a.strict = strict;
  a.clip = clip;
  if (fin_t >= 0) fin_fill(c, a, fin_t, tol, strict);
  HIPCHK(launch_pass<MODE_L>(c->KB, c->data_kind, a, c->chunksH, c->stream));
  if (int rc = enqueue_exchange_after_sweep(c, a, /*with_products=*/false, strict)) return rc;
  return NBMF_OK;
}

int enqueue_finalize(nbmf_ctx* c, int t, double tol, bool loglik_only = false, int strict = 0) {
  // single GPU: per-wave partials + pad correction here; sharded: the exchanged scalar (pad already removed)
  const bool sh = is_sharded(c) && c->ll_ptr;
  const double* ll_src = sh ? c->ll_ptr : c->lossbuf;
  const int n_ll = sh ? 1 : c->chunksH * (int)(c->nA / 16 / wg_strips(c));
  const double pad = sh ? 0.0 : ll_pad_of(c, strict);
  // axis 1: the prior sums were exchanged with the log-likelihood
  const bool prior_x = sh && c->shard_axis == 1;
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, c->stream, ll_src, n_ll, pad,
                     prior_x ? (const double*)(c->sbuf + 1) : c->prior_src,
                     loglik_only ? 0 : (prior_x ? 1 : c->n_prior_src),   // no prior term (and no 0 x NaN) in a pure log-likelihood
                     loglik_only ? 0.0 : c->alpha - 1.0, loglik_only ? 0.0 : c->beta - 1.0,
                     loglik_only ? -1.0 : c->n_obs_global, c->losses_d, t, tol, c->scal, c->flags);
  HIPCHK(hipGetLastError());
  return NBMF_OK;
}

int enqueue_h_update(nbmf_ctx* c) {
  if (c->KS > 1) {
    // the update is elementwise in k: one launch per slice on that slice's rows of H and its slabs
    const size_t per = (size_t)SLICE_K * c->nA;
    const int blocks = (int)(per / 256);
    for (int sl = 0; sl < c->KS; ++sl) {
      const int ks = std::min(SLICE_K, c->k - sl * SLICE_K);
      hipLaunchKernelGGL(h_update_kernel, dim3(blocks), dim3(256), 0, c->stream,
                         (const double*)(c->slabH + (size_t)sl * c->chunksH * per),
                         (const double*)(c->slabH + (size_t)(c->KS + sl) * c->chunksH * per), c->chunksH, per, (long long)c->nA, 0LL,
                         0LL, (long long)c->nA, c->Hn + sl * per, c->HT + sl * per, c->HG + sl * per,
                         c->prior + 2 * (size_t)sl * blocks, ks, SLICE_K, (long long)c->n, (long long)c->nA, c->alpha - 1.0,
                         c->beta - 1.0, c->eps, c->flags, pform_inv12(c));
      HIPCHK(hipGetLastError());
    }
    c->prior_src = c->prior;
    c->n_prior_src = c->n_prior_blocks;
    return NBMF_OK;
  }
  // single GPU or column split: sum the H-pass slabs here, all columns in one launch
  const size_t per = (size_t)c->KP * c->nA;
  hipLaunchKernelGGL(h_update_kernel, dim3(c->n_prior_blocks), dim3(256), 0, c->stream, (const double*)c->slabH,
                     (const double*)(c->slabH + (size_t)c->chunksH * per), c->chunksH, per, (long long)c->nA, 0LL, 0LL,
                     (long long)c->nA, c->Hn, c->HT, c->HG, c->prior, c->k, c->KP, (long long)c->n, (long long)c->nA,
                     c->alpha - 1.0, c->beta - 1.0, c->eps, c->flags, pform_inv12(c));
  HIPCHK(hipGetLastError());
  c->prior_src = c->prior;
  c->n_prior_src = c->n_prior_blocks;
  return NBMF_OK;
}

PassArgs w_pass_args(nbmf_ctx* c) {
  PassArgs a{};
  a.data = sweep_image(c, 1);
  a.mask = c->maskB;
  a.LT = c->HT;
  a.LG = c->HG;
  a.RfT = c->WT;
  a.out1 = c->slabW;
  a.out2 = nullptr;
  a.lossbuf = nullptr;
  a.done = c->flags;
  a.Rb = (int)(c->nA / 16);
  a.Cb = (int)(c->mA / 16);
  a.chunk_start = c->cstartW;
  a.C_alloc = c->mA;
  a.eps = c->eps;
  a.tiny_eps = tiny_a(c) || c->w_free;   // (transform's W steps start from a W that is not on the simplex: the select variant)
  // every entry observed (no mask, or a mask of ones); pad rows of the dimension the W sweep walks (the columns of Y) would
  // count as observed zeros in the two-state variant's column sums: there are none, or they are "ones" in image B's lane
  // masks (mark_pad_rows_b)
  a.full = c->data_kind == DATA_BIN && c->n_obs == (double)c->m * (double)c->n && (c->nA == c->n || c->padB_ones);
  a.ksteps = (c->k + 3) / 4;
  a.kblocks = (c->k + 15) / 16;
  return a;
}

template <int GROUPS>
int launch_w_update(nbmf_ctx* c, const double* q, int chunks, double n_div, int projection) {
  const size_t lds_bytes = sizeof(double) * ((size_t)c->KP + 2) * WU_COLS;
  if (lds_bytes > 65536)
    HIPCHK(hipFuncSetAttribute((const void*)w_update_kernel<GROUPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(w_update_kernel<GROUPS>, dim3((unsigned)(c->mA / WU_COLS)), dim3(WU_COLS * GROUPS), lds_bytes, c->stream, q,
                     chunks, c->Wn, c->WT, c->WG, c->k, c->KP, std::min(c->KP, SLICE_K), (long long)c->m, (long long)c->mA, n_div,
                     c->rowcnt, projection, c->flags);
  HIPCHK(hipGetLastError());
  return NBMF_OK;
}
int enqueue_w_update(nbmf_ctx* c, const double* q, int chunks, double n_div, int projection) {
  // 1024-thread blocks up to two rounds of them (two are resident per CU), 256-thread blocks beyond (w_update_kernel;
  // measured at K = 64: 32768 rows 28 against 30 us, 65536 rows 50 against 38 us)
  return c->mA / WU_COLS <= 4LL * c->cus ? launch_w_update<32>(c, q, chunks, n_div, projection)
                                        : launch_w_update<8>(c, q, chunks, n_div, projection);
}

// Rows of Y split: ordered sum of the H-pass slabs into [P1 | P2 | loglik] at `dst` (natural [KP][nA] twice);
// with slices, one launch per slice on its rows.
int enqueue_reduce_h_all(nbmf_ctx* c, double* dst, hipStream_t st) {
  const int KSK = std::min(c->KP, SLICE_K);
  const size_t per = (size_t)KSK * c->nA, tot = (size_t)c->KP * c->nA;
  const int n_loss = c->chunksH * (int)(c->nA / 16 / wg_strips(c));
  for (int sl = 0; sl < c->KS; ++sl) {
    hipLaunchKernelGGL(reduce_h_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st,
                       (const double*)(c->slabH + (size_t)sl * c->chunksH * per),
                       (const double*)(c->slabH + (size_t)(c->KS + sl) * c->chunksH * per), c->chunksH, (long long)per,
                       (long long)c->nA, 0LL, (long long)c->nA, (long long)c->nA, KSK, dst + sl * per, dst + tot + sl * per,
                       (const double*)c->lossbuf, n_loss, ll_pad_of(c), sl == 0 ? dst + 2 * tot : (double*)nullptr, c->flags, pform_inv12(c));
    HIPCHK(hipGetLastError());
  }
  return NBMF_OK;
}

// H-update of all columns from the summed products at `src` ([P1 | P2], natural), one launch per slice
int enqueue_h_update_from(nbmf_ctx* c, const double* src, hipStream_t st) {
  const int KSK = std::min(c->KP, SLICE_K);
  const size_t per = (size_t)KSK * c->nA, tot = (size_t)c->KP * c->nA;
  const int blocks = (int)(per / 256);
  for (int sl = 0; sl < c->KS; ++sl) {
    const int ks = std::min(KSK, c->k - sl * KSK);
    hipLaunchKernelGGL(h_update_kernel, dim3(blocks), dim3(256), 0, st, src + sl * per, src + tot + sl * per, 1, (size_t)0,
                       (long long)c->nA, 0LL, 0LL, (long long)c->nA, c->Hn + sl * per, c->HT + sl * per, c->HG + sl * per,
                       c->prior + 2 * (size_t)sl * blocks, ks, KSK, (long long)c->n, (long long)c->nA, c->alpha - 1.0,
                       c->beta - 1.0, c->eps, c->flags, /*pform=*/0.0);   // (the exchanged sums are P1, P2 themselves: reduce_h_kernel)
    HIPCHK(hipGetLastError());
  }
  c->prior_src = c->prior;
  c->n_prior_src = c->n_prior_blocks;
  return NBMF_OK;
}

int enqueue_w_step(nbmf_ctx* c, int projection);

// Rows of Y split, peer transport: the H-step exchange is this library's own reduce-scatter over xGMI with
// the H-update fused in (each rank updates 1/R of the columns and broadcasts H', so the update is not
// repeated R times and only K*N instead of 2*K*N doubles come back).  Order on the stream:
//   H-pass -> slab reduction into the arena -> peer_h_kernel (H' staged in every arena, loglik total)
//   -> loss + stop test of iteration it-1 -> peer_h_apply_kernel (skipped once the stop flag is up, so the
//   factors stay those of iteration it-1, as in the single-GPU run) -> W-pass -> W-update.
int enqueue_iteration_rows_peer(nbmf_ctx* c, int it, double tol) {
  const size_t per = (size_t)c->KP * c->nA;
  PassArgs a{};
  a.data = sweep_image(c, 0);
  a.mask = c->maskA;
  a.LT = c->WT;
  a.LG = c->WG;
  a.RfT = c->HT;
  a.out1 = c->slabH;
  a.out2 = c->slabH + (size_t)c->chunksH * per;
  a.lossbuf = c->lossbuf;
  a.done = c->flags;
  a.Rb = (int)(c->mA / 16);
  a.Cb = (int)(c->nA / 16);
  a.chunk_start = c->cstartH;
  a.C_alloc = c->nA;
  a.eps = c->eps;
  a.tiny_eps = tiny_a(c);
  a.ksteps = (c->k + 3) / 4;
  a.kblocks = (c->k + 15) / 16;
  {
    EvScope ev(c, 0, true, /*attach=*/c->KS == 1);
    if (c->KS > 1) {
      if (int rc = enqueue_a_sweeps_sliced(c, true, 0, 0)) return rc;
    } else {
      HIPCHK(launch_pass<MODE_H>(c->KB, c->data_kind, a, c->chunksH, c->stream));
    }
  }
  double* X = c->arena;   // [P1 (KP x nA) | P2 (KP x nA) | loglik], natural order
  const unsigned long long hs = ++c->hseq;
  const int slots_per_panel = PEER_H_WGS * c->pv.nranks;
  const long long offPR_now = c->offPR + (long long)(hs & 1) * 2 * 2 * PEER_H_WGS * PEER_MAX_RANKS;
  double* ll_slot = c->scal + 4;
  // the loss and stop test of iteration it-1 ride in the exchange kernel (its first workgroup): they read the
  // prior sums the PREVIOUS exchange left in the arena
  const long long offPR_prev = c->prior_src ? (long long)(c->prior_src - c->arena) : 0;
  const bool prev_in_arena = c->prior_src >= c->arena && c->prior_src < c->arena + c->arena_doubles;
  const int fin_t = (it > 0 && prev_in_arena) ? it - 1 : -1;
  const int n_loss = c->chunksH * (a.Cb / wg_strips(c));
  // One column panel [pc0[p], pc0[p+1]) of the H-step on stream `st`: ordered slab sum into the arena, then the
  // fused reduce-scatter + Beta-MAP update + broadcast of this rank's slice of the panel (flag slot p).
  auto exchange_panel = [&](int p, hipStream_t st, unsigned long long e) -> int {
    const long long c0 = c->pc0[p], wp = c->pc0[p + 1] - c0;
    if (c->KS > 1) {
      if (int rc = enqueue_reduce_h_all(c, X, st)) return rc;   // (slices: one panel, see comm_finish_init)
    } else {
      hipLaunchKernelGGL(reduce_h_kernel, dim3((unsigned)(((long long)c->KP * wp + 255) / 256)), dim3(256), 0, st,
                         (const double*)a.out1, (const double*)a.out2, c->chunksH, (long long)per, (long long)c->nA, c0, wp,
                         (long long)c->nA, c->KP, X + c0, X + per + c0, (const double*)c->lossbuf, n_loss, ll_pad_of(c),
                         p == 0 ? X + 2 * per : (double*)nullptr, c->flags, pform_inv12(c));
      HIPCHK(hipGetLastError());
    }
    PeerView pv = c->pv;
    pv.fbase = p * PF_SLOT;
    hipLaunchKernelGGL(peer_h_kernel, dim3(PEER_H_WGS), dim3(256), 0, st, pv, e, (long long)per, p == 0 ? (long long)(2 * per) : -1LL,
                       c->offHX, offPR_now, p * slots_per_panel + PEER_H_WGS * c->pv.rank, c->psl_c0[p], c->psl_wp[p],
                       (const double*)c->Hn, c->k, c->KP, (long long)c->n, (long long)c->nA, c->alpha - 1.0, c->beta - 1.0, c->eps,
                       ll_slot, c->flags, p == 0 ? fin_t : -1, offPR_prev, c->n_prior_src, c->n_obs_global, c->losses_d, tol,
                       c->scal, /*pform=*/0.0);   // (every rank's arena holds P1, P2 themselves: reduce_h_kernel)
    HIPCHK(hipGetLastError());
    return NBMF_OK;
  };
  auto apply_panel = [&](int p, hipStream_t st, unsigned long long e) -> int {
    PeerView pv = c->pv;
    pv.fbase = p * PF_SLOT;
    hipLaunchKernelGGL(peer_h_apply_kernel, dim3(PEER_H_WGS), dim3(256), 0, st, pv, e, c->offHX, c->Hn, c->HT, c->HG, c->KP,
                       std::min(c->KP, SLICE_K), (long long)c->nA, c->pc0[p], c->pc0[p + 1] - c->pc0[p], c->flags);
    HIPCHK(hipGetLastError());
    return NBMF_OK;
  };
  const bool two = c->npanel == 2 && c->stream2;
  hipStream_t s0 = c->stream, s1 = two ? c->stream2 : c->stream;
  const unsigned long long e0 = ++c->epoch, e1 = two ? ++c->epoch : 0;
  // ---- panel 0 (with the log-likelihood, the loss and the stop test of iteration it-1) on the main stream
  if (int rc = exchange_panel(0, s0, e0)) return rc;
  c->ll_ptr = ll_slot;
  if (it > 0 && fin_t < 0)
    if (int rc = enqueue_finalize(c, it - 1, tol)) return rc;   // (first iteration after nbmf_set_factors: local prior sums)
  c->prior_src = c->arena + offPR_now;
  c->n_prior_src = c->npanel * slots_per_panel;
  if (!two) {
    if (int rc = apply_panel(0, s0, e0)) return rc;
    return enqueue_w_step(c, c->projection);   // the rows of W are local: no exchange in the W-step
  }
  // ---- two panels: panel 1's exchange travels on the side stream while panel 0 is applied and the W-pass sweeps
  //      the chunks that read only panel 0's columns of H' (it is chunked over exactly that index); panel 1's H'
  //      is applied behind the stop test and feeds the remaining chunks there.
  // (panel 1's exchange kernel starts behind panel 0's, which settles the stop flag of iteration it-1: every rank
  //  must see the same flag when its kernel decides whether to take part, or the others wait for it in vain)
  HIPCHK(hipEventRecord(c->evF, s0));
  HIPCHK(hipStreamWaitEvent(s1, c->evF, 0));
  if (int rc = exchange_panel(1, s1, e1)) return rc;
  if (int rc = apply_panel(1, s1, e1)) return rc;
  PassArgs w = w_pass_args(c);
  {
    PassArgs w1 = w;
    w1.chunk0 = c->wsplit;
    HIPCHK(launch_pass<MODE_W>(c->KB, c->data_kind, w1, c->chunksW - c->wsplit, s1));
    HIPCHK(hipEventRecord(c->ev1, s1));
  }
  if (int rc = apply_panel(0, s0, e0)) return rc;
  {
    EvScope ev(c, 1);   // until both halves are done (includes any wait for panel 1's exchange)
    HIPCHK(launch_pass<MODE_W>(c->KB, c->data_kind, w, c->wsplit, s0));
    HIPCHK(hipStreamWaitEvent(s0, c->ev1, 0));
  }
  return enqueue_w_update(c, c->slabW, c->chunksW, (double)c->n, c->projection);
}

// One whole iteration when the ROWS of Y are split over the ranks.  The exchange [P1 | P2 | loglik] is
// cut into (at most) two column panels; panel 1's reduction + all-reduce + H-update run on stream2 while
// stream 1 already updates panel 0 and sweeps the W-pass chunks that only read panel 0's columns of H'
// (the W-pass is chunked over exactly that index).  With the host transport everything stays on one
// stream (it synchronises anyway); the arithmetic is the same.
int enqueue_iteration_rows(nbmf_ctx* c, int it, double tol) {
  if (c->peer) return enqueue_iteration_rows_peer(c, it, tol);
  if (c->KS > 1) {
    // slices: one exchange of the whole [P1 | P2 | loglik], no panels
    const size_t tot = (size_t)c->KP * c->nA;
    {
      EvScope ev(c, 0);
      if (int rc = enqueue_a_sweeps_sliced(c, true, 0, 0)) return rc;
    }
    if (int rc = enqueue_reduce_h_all(c, c->Pbuf, c->stream)) return rc;
    if (int rc = all_reduce_inplace(c, c->Pbuf, 2 * tot + 1)) return rc;
    c->ll_ptr = c->Pbuf + 2 * tot;
    if (it > 0)
      if (int rc = enqueue_finalize(c, it - 1, tol)) return rc;
    if (int rc = enqueue_h_update_from(c, c->Pbuf, c->stream)) return rc;
    return enqueue_w_step(c, c->projection);
  }
  hipStream_t s0 = c->stream;
  const bool two_streams = c->npanel == 2 && c->comm && c->stream2;
  hipStream_t s1 = two_streams ? c->stream2 : s0;
  const size_t per = (size_t)c->KP * c->nA;
  // ---- H-pass (all columns)
  PassArgs a{};
  a.data = sweep_image(c, 0);
  a.mask = c->maskA;
  a.LT = c->WT;
  a.LG = c->WG;
  a.RfT = c->HT;
  a.out1 = c->slabH;
  a.out2 = c->slabH + (size_t)c->chunksH * per;
  a.lossbuf = c->lossbuf;
  a.done = c->flags;
  a.Rb = (int)(c->mA / 16);
  a.Cb = (int)(c->nA / 16);
  a.chunk_start = c->cstartH;
  a.C_alloc = c->nA;
  a.eps = c->eps;
  a.tiny_eps = tiny_a(c);
  a.ksteps = (c->k + 3) / 4;
  a.kblocks = (c->k + 15) / 16;
  {
    EvScope ev(c, 0, true, /*attach=*/true);
    HIPCHK(launch_pass<MODE_H>(c->KB, c->data_kind, a, c->chunksH, s0));
  }
  if (two_streams) HIPCHK(hipEventRecord(c->evH, s0));
  const int n_loss = c->chunksH * (a.Cb / wg_strips(c));
  auto reduce_panel = [&](int p, hipStream_t st) -> int {
    const long long c0 = c->pc0[p], wp = c->pc0[p + 1] - c0;
    double* d1 = c->Pbuf + c->pbase[p];
    double* d2 = d1 + (size_t)c->KP * wp;
    hipLaunchKernelGGL(reduce_h_kernel, dim3((unsigned)(((long long)c->KP * wp + 255) / 256)), dim3(256), 0, st,
                       (const double*)a.out1, (const double*)a.out2, c->chunksH, (long long)per, (long long)c->nA, c0, wp, wp,
                       c->KP, d1, d2, (const double*)c->lossbuf, n_loss, ll_pad_of(c),
                       p == 0 ? c->Pbuf + c->ll_index : (double*)nullptr, c->flags, pform_inv12(c));
    HIPCHK(hipGetLastError());
    return all_reduce_inplace(c, d1, 2 * (size_t)c->KP * wp + (p == 0 ? 1 : 0), st);
  };
  auto update_panel = [&](int p, hipStream_t st) -> int {
    const long long c0 = c->pc0[p], wp = c->pc0[p + 1] - c0;
    const double* d1 = c->Pbuf + c->pbase[p];
    const unsigned blk0 = (unsigned)((long long)c->KP * c0 / 256);
    hipLaunchKernelGGL(h_update_kernel, dim3((unsigned)((long long)c->KP * wp / 256)), dim3(256), 0, st, d1,
                       d1 + (size_t)c->KP * wp, 1, (size_t)0, wp, c0, c0, wp, c->Hn, c->HT, c->HG, c->prior + 2 * (size_t)blk0,
                       c->k, c->KP, (long long)c->n, (long long)c->nA, c->alpha - 1.0, c->beta - 1.0, c->eps, c->flags, /*pform=*/0.0);
    HIPCHK(hipGetLastError());
    c->prior_src = c->prior;
    c->n_prior_src = c->n_prior_blocks;
    return NBMF_OK;
  };
  // ---- panel 0 (+ loglik) on the main stream; the loss of iteration it-1 and its stop test
  if (int rc = reduce_panel(0, s0)) return rc;
  c->ll_ptr = c->Pbuf + c->ll_index;
  if (it > 0)
    if (int rc = enqueue_finalize(c, it - 1, tol)) return rc;
  if (two_streams) HIPCHK(hipEventRecord(c->evF, s0));
  // ---- panel 1 on the side stream: behind the H-pass, its update behind the stop test, then straight
  //      into the W-pass chunks that read panel 1's columns (they run beside the other half on the GPU)
  PassArgs w = w_pass_args(c);
  const int first = c->npanel == 2 ? c->wsplit : c->chunksW;
  if (c->npanel == 2) {
    if (two_streams) HIPCHK(hipStreamWaitEvent(s1, c->evH, 0));
    if (int rc = reduce_panel(1, s1)) return rc;
    if (two_streams) HIPCHK(hipStreamWaitEvent(s1, c->evF, 0));
    if (int rc = update_panel(1, s1)) return rc;
    if (two_streams) {
      PassArgs w1 = w;
      w1.chunk0 = first;
      HIPCHK(launch_pass<MODE_W>(c->KB, c->data_kind, w1, c->chunksW - first, s1));
      HIPCHK(hipEventRecord(c->ev1, s1));
    }
  }
  if (int rc = update_panel(0, s0)) return rc;
  // ---- W-pass: the chunks over panel 0's columns on the main stream
  {
    EvScope ev(c, 1);   // with two streams: until both halves are done (includes any wait for panel 1's exchange)
    HIPCHK(launch_pass<MODE_W>(c->KB, c->data_kind, w, first, s0));
    if (c->npanel == 2) {
      if (two_streams) {
        HIPCHK(hipStreamWaitEvent(s0, c->ev1, 0));
      } else {
        w.chunk0 = first;
        HIPCHK(launch_pass<MODE_W>(c->KB, c->data_kind, w, c->chunksW - first, s0));
      }
    }
  }
  return enqueue_w_update(c, c->slabW, c->chunksW, (double)c->n, c->projection);
}

int enqueue_w_step(nbmf_ctx* c, int projection) {
  if (c->KS > 1) {
    {
      EvScope ev(c, 1);
      if (int rc = enqueue_theta(c, 1)) return rc;
      for (int step = 0; step < c->KS; ++step) {   // the last slice first: it completes and stores Theta'
        const int sl = step == 0 ? c->KS - 1 : step - 1;
        PassArgs w = w_pass_args(c);
        w.LT = c->HT + (size_t)sl * SLICE_K * c->nA;
        w.LG = c->HG + (size_t)sl * SLICE_K * c->nA;
        w.RfT = c->WT + (size_t)sl * SLICE_K * c->mA;
        w.out1 = c->slabW + (size_t)sl * c->chunksW * SLICE_K * c->mA;
        w.theta = c->theta;
        if (step == 0)
          HIPCHK((launch_pass_slice<MODE_W, 2>(c->data_kind, w, c->chunksW, c->stream)));
        else
          HIPCHK((launch_pass_slice<MODE_W, 1>(c->data_kind, w, c->chunksW, c->stream)));
      }
    }
    if (is_sharded(c) && c->shard_axis == 1) {
      // columns of Y split: sum every slice's slabs into its rows of Qbuf, one exchange of K_pad x m
      const long long per = (long long)SLICE_K * c->mA;
      for (int sl = 0; sl < c->KS; ++sl) {
        hipLaunchKernelGGL(reduce_w_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, c->stream,
                           (const double*)(c->slabW + (size_t)sl * c->chunksW * per), c->Qbuf + sl * per, c->chunksW, per, c->flags);
        HIPCHK(hipGetLastError());
      }
      if (int rc = all_reduce_inplace(c, c->Qbuf, (size_t)c->KP * c->mA)) return rc;
      return enqueue_w_update(c, c->Qbuf, 1, c->n_div_global, projection);
    }
    return enqueue_w_update(c, c->slabW, c->chunksW, (double)c->n, projection);
  }
  PassArgs a = w_pass_args(c);
  {
    EvScope ev(c, 1, true, /*attach=*/true);
    HIPCHK(launch_pass<MODE_W>(c->KB, c->data_kind, a, c->chunksW, c->stream));
  }
  const double* q = c->slabW;
  int chunks = c->chunksW;
  double n_div = (double)c->n;
  if (is_sharded(c) && c->shard_axis == 1) {
    // columns of Y are split: the bracket of _solver.py:53 is a sum over ranks -> one all-reduce of K x m
    const long long per = (long long)c->KP * c->mA;
    hipLaunchKernelGGL(reduce_w_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, c->stream,
                       (const double*)c->slabW, c->Qbuf, c->chunksW, per, c->flags);
    HIPCHK(hipGetLastError());
    if (int rc = all_reduce_inplace(c, c->Qbuf, (size_t)per)) return rc;
    q = c->Qbuf;
    chunks = 1;
    n_div = c->n_div_global;
  }
  return enqueue_w_update(c, q, chunks, n_div, projection);
}

// ---- small problems: the fit in one persistent launch (nbmf_small_kernel.inc) ------------------------------
// natural padded [KP][lenA] -> T and G images (restoring the factors after an abandoned persistent run)
__global__ void expand_factor_kernel(const double* __restrict__ Fn, double* __restrict__ FT, double* __restrict__ FG, int KP,
                                     int KS, long long lenA) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)KP * lenA) return;
  const long long k = idx / lenA, x = idx % lenA;
  FT[t_index(k, x, KS, lenA)] = Fn[idx];
  FG[g_index(k, x, KS, lenA)] = Fn[idx];
}

template <int KB, bool SPLIT, bool BATCH>
const void* small_ptr(int data_kind) {
  switch (data_kind) {
    case DATA_BIN: return (const void*)small_fit_kernel<KB, DATA_BIN, SPLIT, BATCH>;
    case DATA_F64: return (const void*)small_fit_kernel<KB, DATA_F64, SPLIT, BATCH>;
    case DATA_F64M: return (const void*)small_fit_kernel<KB, DATA_F64M, SPLIT, BATCH>;
  }
  return nullptr;
}
const void* small_kernel_for(const nbmf_ctx* c, bool split, bool batch);

// How many workgroups a strip of the H-step (sweep of Rbe blocks, Cbe strips) or of the W-step is split over: aim at
// two tiles per wave, stay within the CUs, at most 8 parts (one partner hand-off costs ~2 us: not worth it for
// shorter sweeps).
constexpr int SMALL_MAX_WGS = 256;   // workgroups a persistent run may have: the hand-off words, flags and partial buffers are sized for it
int small_parts(const nbmf_ctx* c, long long sweep_blocks, long long strips) {
  const int NW = sm_waves(c->KB);
  long long pp = (sweep_blocks + 2 * NW - 1) / (2 * NW);
  pp = std::min<long long>(pp, std::min<long long>(8, std::min(c->cus, SMALL_MAX_WGS) / std::max<long long>(1, strips)));
  return (int)std::max<long long>(1, pp);
}

// Is this run one for the persistent kernel?  One GPU, at most 32 components (the 16-wave workgroups leave 128
// registers per lane), no per-launch instrumentation asked for, every workgroup on a CU of its own, at most SM_TPW
// tiles per wave and phase, at most 128 H-strips (the loss is summed by one wave, two strips per lane) -- and not
// more than NBMF_SMALL_TILES tiles (default 32768, which the other bounds imply anyway: at 4080 x 2040, K = 16 the
// single launch still runs 1.4x the five-kernel path; larger problems belong to the LDS-staged pass kernels).
bool small_eligible(const nbmf_ctx* c, int cus) {
  // (a progress callback does not change the engine: the run takes milliseconds, its losses are reported right after
  //  it, in the same batches -- so a verbose fit and a silent one give the same bits)
  if (is_sharded(c) || c->KS != 1 || c->KB > 2 || c->timing) return false;
  cus = std::min(cus, SMALL_MAX_WGS);   // a device reporting more CUs than the buffers were sized for uses that many of them
  if (const char* e = getenv("NBMF_PERSISTENT"))
    if (atoi(e) == 0) return false;
  const long long Rbe = (c->m + 15) / 16, Cbe = (c->n + 15) / 16;
  long long max_tiles = 32768;
  if (const char* e = getenv("NBMF_SMALL_TILES")) max_tiles = atoll(e);
  const int NW = sm_waves(c->KB), PH = small_parts(c, Rbe, Cbe), PW = small_parts(c, Cbe, Rbe);
  const bool fits = (Rbe + (long long)PH * NW - 1) / ((long long)PH * NW) <= SM_TPW && (Cbe + (long long)PW * NW - 1) / ((long long)PW * NW) <= SM_TPW;
  if (!(fits && Cbe <= 128 && Cbe * PH <= cus && Rbe * PW <= cus && Rbe * Cbe <= max_tiles)) return false;
  // the problem qualifies.  A context whose persistent kernel once gave up keeps to the launches -- counted, so that a
  // test which names the single-launch engine can tell that it did not run (nbmf_engine_stats)
  if (c->small.disabled) {
    g_engine_persistent_declined.fetch_add(1, std::memory_order_relaxed);
    return false;
  }
  return true;
}

// Admission of persistent kernels, per device and process-wide: a persistent kernel needs ALL its workgroups
// resident (one per CU) or its first barrier never completes.  Several host threads may run small fits on one GPU
// at once (experiments.perplexity_grid(concurrency=...)); launched together, their kernels would each get a share
// of the CUs and wait for the rest forever.  So a run reserves its CUs first and waits (milliseconds: these runs
// are short) while the ones in flight hold too many.
struct SmallAdmission {
  std::mutex mu;
  std::condition_variable cv;
  int busy[64] = {0};
} g_small_admission;

struct SmallReservation {
  int device, n;
  SmallReservation(int device_, int n_, int capacity) : device(device_ & 63), n(n_) {
    std::unique_lock<std::mutex> lk(g_small_admission.mu);
    g_small_admission.cv.wait(lk, [&] { return g_small_admission.busy[device] + n <= capacity || g_small_admission.busy[device] == 0; });
    g_small_admission.busy[device] += n;
  }
  ~SmallReservation() {
    {
      std::lock_guard<std::mutex> lk(g_small_admission.mu);
      g_small_admission.busy[device] -= n;
    }
    g_small_admission.cv.notify_all();
  }
};

int small_prepare(nbmf_ctx* c) {
  if (c->small.ready) return NBMF_OK;
  // one allocation carved up (a fit of a small problem is over in milliseconds: a dozen hipMallocs would show)
  const size_t fw = (size_t)c->KP * c->mA * sizeof(double), fh = (size_t)c->KP * c->nA * sizeof(double);
  const size_t g_max = 16 * SM_TPW;
  const size_t sync_b = round_up(sizeof(unsigned long long) * (SMALL_MAX_WGS + 8), 256), ll_b = round_up(sizeof(double) * 2 * g_max, 256),
               pr_b = sizeof(double) * 3 * g_max * 8 * 2, res_b = 256;
  const size_t g_all = SMALL_MAX_WGS;   // most workgroups a run can have (one per CU; small_eligible holds runs to it)
  const size_t pb_b = sizeof(double) * g_all * 2 * 16 * (size_t)c->KP, pl_b = sizeof(double) * g_all, pf_b = sizeof(unsigned long long) * g_all;
  const size_t total = 4 * fw + 4 * fh + sync_b + ll_b + pr_b + res_b + pb_b + pl_b + pf_b;
  char* base = nullptr;
  HIPCHK(dmalloc(&base, total));
  HIPCHK(hipMemsetAsync(base, 0, total, c->stream));   // (pad strips of the second set of images are never written: they must read as zero)
  c->small.slab = base;
  char* p = base;
  for (double** q : {&c->small.Wn, &c->small.WT, &c->small.WG, &c->small.snapW}) {
    *q = (double*)p;
    p += fw;
  }
  for (double** q : {&c->small.Hn, &c->small.HT, &c->small.HG, &c->small.snapH}) {
    *q = (double*)p;
    p += fh;
  }
  c->small.sync = (unsigned long long*)p;   // [strip] epoch words + abort word
  p += sync_b;
  c->small.ll_part = (double*)p;            // [parity][strip]
  p += ll_b;
  c->small.prior_part = (double*)p;         // [iteration mod 3][strip][updating wave] (sized for pairs: the second half is unused)
  p += pr_b;
  c->small.result = (int*)p;
  p += res_b;
  c->small.part_buf = (double*)p;           // [workgroup][2][16 KP]
  p += pb_b;
  c->small.part_ll = (double*)p;
  p += pl_b;
  c->small.part_flag = (unsigned long long*)p;
  c->small.ready = true;
  return NBMF_OK;
}

const void* small_kernel_for(const nbmf_ctx* c, bool split, bool batch) {
  if (c->KB == 1)
    return batch ? (split ? small_ptr<1, true, true>(c->data_kind) : small_ptr<1, false, true>(c->data_kind))
                 : (split ? small_ptr<1, true, false>(c->data_kind) : small_ptr<1, false, false>(c->data_kind));
  return batch ? (split ? small_ptr<2, true, true>(c->data_kind) : small_ptr<2, false, true>(c->data_kind))
               : (split ? small_ptr<2, true, false>(c->data_kind) : small_ptr<2, false, false>(c->data_kind));
}

// Everything in SmallArgs that a run shares with every other run on this context's data: the packed images, the
// geometry, how strips are split over workgroups, the hyper-parameters that are not per problem.
void small_common_args(const nbmf_ctx* c, int max_iter, double tol, SmallArgs* out) {
  SmallArgs& a = *out;
  a.dataA = c->dataA;
  a.maskA = c->maskA;
  a.dataB = c->dataB;
  a.maskB = c->maskB;
  a.rowcnt = c->rowcnt;
  a.m = c->m; a.n = c->n; a.mA = c->mA; a.nA = c->nA;
  a.K = c->k;
  a.RbA = (int)(c->mA / 16);
  a.CbA = (int)(c->nA / 16);
  a.Rbe = (int)((c->m + 15) / 16);
  a.Cbe = (int)((c->n + 15) / 16);
  // strips whose sweep is long are split over several workgroups, a CU each (small_parts)
  a.PH = small_parts(c, a.Rbe, a.Cbe);
  a.PW = small_parts(c, a.Cbe, a.Rbe);
  if (const char* e = getenv("NBMF_SMALL_PARTS")) {   // "ph,pw" (experiments; must still satisfy small_eligible's bounds)
    int x = 0, y = 0;
    const int cap = std::min(c->cus, SMALL_MAX_WGS);
    if (sscanf(e, "%d,%d", &x, &y) == 2 && x >= a.PH && y >= a.PW && x <= 8 && y <= 8 && a.Cbe * x <= cap && a.Rbe * y <= cap) {
      a.PH = x;
      a.PW = y;
    }
  }
  a.G = std::max(a.Cbe * a.PH, a.Rbe * a.PW);
  a.max_iter = max_iter;
  a.projection = c->projection;
  a.tiny_eps = c->eps < 1e-70;
  a.tol = tol;
  a.eps = c->eps;
  a.am1 = c->alpha - 1.0;
  a.bm1 = c->beta - 1.0;
  a.n_obs = c->n_obs_global;
  a.n_div = (double)c->n;
  a.ll_pad = (256.0 * a.Rbe * a.Cbe - (double)c->m * (double)c->n) * log(1.0 + c->eps);
  double ms = 2000.0;
  if (const char* e = getenv("NBMF_SMALL_TIMEOUT_MS")) ms = std::max(1.0, atof(e));
  a.timeout = (unsigned long long)(ms * 1e5);
  if (const char* e = getenv("NBMF_SMALL_FENCED")) a.fenced = atoi(e) != 0;   // release / acquire around every hand-off (nbmf_small_kernel.inc: small_arrive)
}

// END-OF-RUN GUARD of the single-launch engine (round 6).  Its hand-offs are the guide's measured sc1 form, not the memory
// model's release / acquire (which costs more than a phase: DESIGN.md 4.4): a stale read would be a silent wrong answer.
// What CAN be checked from outside at the price of two small launches: the last loss the persistent kernel reported was
// formed from the factors as its workgroups SAW them across the final hand-offs; the launch-per-kernel engine recomputes
// the loss of the factors as they ARE in memory (a kernel boundary: no hand-off involved).  The two engines agree to
// 1e-12 relative when nothing went wrong (different summation orders; tests/test_gpu_parity.py holds them to that), a stale
// factor entry in the last sweep moves the loss by far more unless the fit has converged to that very precision.  A
// mismatch is treated like an abandoned barrier: the run is redone by the launches from the snapshot and counted
// (nbmf_engine_stats: gave_up).  NBMF_SMALL_GUARD=0 switches the check off; NBMF_SMALL_GUARD_FAULT=1 (tests) makes it trip.
int loss_by_launches(nbmf_ctx* c, double* out);
bool small_guard_on() {
  static const bool on = !(getenv("NBMF_SMALL_GUARD") && atoi(getenv("NBMF_SMALL_GUARD")) == 0);
  return on;
}
int small_guard(nbmf_ctx* c, double reported, bool* ok) {
  double chk = 0.0;
  if (int rc = loss_by_launches(c, &chk)) return rc;
  if (getenv("NBMF_SMALL_GUARD_FAULT")) chk += 1e-6 * std::fabs(chk) + 1e-300;
  *ok = (std::isnan(chk) && std::isnan(reported)) || chk == reported ||
        std::fabs(chk - reported) <= 1e-12 * std::max(std::fabs(chk), std::fabs(reported));
  if (!*ok && getenv("NBMF_DEBUG"))
    fprintf(stderr, "[nbmf] single-launch guard: reported loss %.17g, recomputed %.17g: redone by the launches\n", reported, chk);
  return NBMF_OK;
}

// Returns NBMF_OK with *handled = true when the run is complete (losses, n_iter filled, factors in the context's
// buffers); *handled = false means "use the five-kernel path" (factors restored to their state at entry).
int run_small(nbmf_ctx* c, int max_iter, double tol, double* losses, int* n_iter, bool* handled) {
  *handled = false;
  if (!small_eligible(c, c->cus)) return NBMF_OK;
  if (int rc = small_prepare(c)) return rc;
  auto& w = c->small;
  const size_t fw = (size_t)c->KP * c->mA * sizeof(double), fh = (size_t)c->KP * c->nA * sizeof(double);
  SmallArgs a{};
  small_common_args(c, max_iter, tol, &a);
  a.Wn[0] = w.Wn; a.WT[0] = w.WT; a.WG[0] = w.WG; a.Hn[0] = w.Hn; a.HT[0] = w.HT; a.HG[0] = w.HG;
  a.Wn[1] = c->Wn; a.WT[1] = c->WT; a.WG[1] = c->WG; a.Hn[1] = c->Hn; a.HT[1] = c->HT; a.HG[1] = c->HG;
  a.sync = w.sync;
  a.ll_part = w.ll_part;
  a.prior_part = w.prior_part;
  a.losses = c->losses_d;
  a.result = w.result;
  a.part_buf = w.part_buf;
  a.part_ll = w.part_ll;
  a.part_flag = w.part_flag;
  const bool split = a.PH > 1 || a.PW > 1;
  const void* f = small_kernel_for(c, split, false);
  if (!f) return NBMF_OK;
  const int NW = sm_waves(c->KB);
  const size_t lds_bytes = sizeof(double) * ((size_t)NW * 2 * c->KB * 4 * 64 + NW * 16 + 64 + 2 * c->KP * 16) + LOG_TABLE_BYTES;
  HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  // snapshot for the fall-back, hand-off words cleared
  HIPCHK(hipMemcpyAsync(w.snapW, c->Wn, fw, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(w.snapH, c->Hn, fh, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemsetAsync(w.sync, 0, sizeof(unsigned long long) * (a.G + 1), c->stream));
  HIPCHK(hipMemsetAsync(w.part_flag, 0, sizeof(unsigned long long) * SMALL_MAX_WGS, c->stream));
  HIPCHK(hipMemsetAsync(w.result, 0, sizeof(int) * 4, c->stream));
  if (getenv("NBMF_SMALL_FORCE_ABORT")) {   // tests: raise the abort word up front, so that the fall-back runs
    const unsigned long long one = 1;
    HIPCHK(hipMemcpyAsync(w.sync + a.G, &one, sizeof one, hipMemcpyHostToDevice, c->stream));
  }
  // Whichever way this function returns from here on -- every HIPCHK below is an early return -- the stream is
  // drained BEFORE the diagnostic buffers go back to the pool and BEFORE the CU reservation is released: a persistent
  // kernel that is still resident must not see its buffers reused, nor another thread's kernel admitted beside it.
  // (Declaration order: destructors run in reverse, the synchronisation first.)
  unsigned long long *dbg = nullptr, *place = nullptr;
  struct DiagGuard {
    unsigned long long *&a, *&b;
    ~DiagGuard() {
      if (a) dfree(a);
      if (b) dfree(b);
    }
  } diag_guard{dbg, place};
  if (getenv("NBMF_SMALL_DEBUG")) {
    HIPCHK(dmalloc(&dbg, sizeof(unsigned long long) * 64 * 16));
    HIPCHK(hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * 64 * 16, c->stream));
    a.dbg = dbg;
  }
  if (getenv("NBMF_SMALL_PLACEMENT")) {
    HIPCHK(dmalloc(&place, sizeof(unsigned long long) * a.G));
    a.place = place;
  }
  void* params[] = {&a};
  SmallReservation cus_held(c->device, a.G, std::min(c->cus, SMALL_MAX_WGS));
  struct DrainGuard {
    hipStream_t st;
    ~DrainGuard() { hipStreamSynchronize(st); }
  } drain_guard{c->stream};
  HIPCHK(hipLaunchKernel(f, dim3(a.G), dim3(64 * NW), params, lds_bytes, c->stream));
  if (place) {   // where the dispatcher put the workgroups: XCD . shader engine . CU
    std::vector<unsigned long long> h(a.G);
    HIPCHK(hipMemcpyAsync(h.data(), place, sizeof(unsigned long long) * a.G, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    fprintf(stderr, "[nbmf] placement (xcd.se.cu):");
    for (int i = 0; i < a.G; ++i)
      fprintf(stderr, " %u.%u.%u", (unsigned)(h[i] >> 32) & 15u, (unsigned)(h[i] >> 13) & 7u, (unsigned)(h[i] >> 8) & 15u);
    fprintf(stderr, "\n");
  }
  if (dbg) {   // per-phase wall clock of workgroup 0 (10 ns ticks): tiles | update | barrier | tiles | update+loss | barrier
    unsigned long long h[64 * 16];
    HIPCHK(hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double acc[8] = {0}, fine[7] = {0};
    int cnt = 0;
    for (int t = 8; t < 63 && t + 1 < max_iter; ++t) {
      if (!h[t * 16 + 7] || !h[(t + 1) * 16]) continue;
      for (int p = 0; p < 7; ++p) acc[p] += (double)(h[t * 16 + p + 1] - h[t * 16 + p]) * 0.01;
      acc[7] += (double)(h[(t + 1) * 16] - h[t * 16 + 7]) * 0.01;
      fine[0] += (double)(h[t * 16 + 8] - h[t * 16 + 3]) * 0.01;    // barrier exit -> operand loads issued
      fine[1] += (double)(h[t * 16 + 9] - h[t * 16 + 8]) * 0.01;    // -> operands landed
      fine[2] += (double)(h[t * 16 + 10] - h[t * 16 + 9]) * 0.01;   // -> first tile's arithmetic done
      fine[3] += (double)(h[t * 16 + 11] - h[t * 16 + 10]) * 0.01;  // -> second tile's
      fine[4] += (double)(h[t * 16 + 12] - h[t * 16 + 1]) * 0.01;   // H: tiles done -> all waves' partials in LDS
      fine[5] += (double)(h[t * 16 + 13] - h[t * 16 + 12]) * 0.01;  // -> update computed, stores issued
      fine[6] += (double)(h[t * 16 + 2] - h[t * 16 + 13]) * 0.01;   // -> prior sums published
      ++cnt;
    }
    if (cnt)
      fprintf(stderr, "[nbmf]   W tiles of wave 0 in detail: issue %.2f | operands land %.2f | tile A %.2f | tile B %.2f;  H reduce+update: "
                      "wait for all waves %.2f | update %.2f | prior sums %.2f\n", fine[0] / cnt, fine[1] / cnt, fine[2] / cnt, fine[3] / cnt,
              fine[4] / cnt, fine[5] / cnt, fine[6] / cnt);
    if (h[15] && h[63 * 16 + 15] > h[15])
      fprintf(stderr, "[nbmf]   on the device from the first instruction to the last: %.3f ms\n", (double)(h[63 * 16 + 15] - h[15]) * 1e-5);
    if (h[8 * 16 + 14] && h[62 * 16 + 14] && h[62 * 16] > h[8 * 16])
      fprintf(stderr, "[nbmf]   shader clock over iterations 8..62: %.0f MHz (%llu cycles in %.2f us)\n",
              (double)(h[62 * 16 + 14] - h[8 * 16 + 14]) / ((double)(h[62 * 16] - h[8 * 16]) * 0.01),
              (unsigned long long)(h[62 * 16 + 14] - h[8 * 16 + 14]), (double)(h[62 * 16] - h[8 * 16]) * 0.01);
    if (cnt)
      fprintf(stderr, "[nbmf] persistent fit, workgroup 0, us per iteration part (mean of %d): H tiles %.2f | H reduce+update %.2f | barrier %.2f | "
                      "W tiles %.2f | W reduce+update %.2f | loss %.2f | barrier %.2f | loop %.2f\n", cnt, acc[0] / cnt, acc[1] / cnt,
              acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt, acc[6] / cnt, acc[7] / cnt);
  }
  int res[4] = {0, 0, 0, 0};
  unsigned long long abort_word = 0;
  HIPCHK(hipMemcpyAsync(res, w.result, sizeof res, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(&abort_word, w.sync + a.G, sizeof abort_word, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(stream_wait_spin(c->stream));
  ++w.runs;
  auto back_to_entry = [&](bool disable) -> int {
    ++w.aborted;
    g_engine_persistent_aborted.fetch_add(1, std::memory_order_relaxed);
    if (disable) w.disabled = true;
    HIPCHK(hipMemcpyAsync(c->Wn, w.snapW, fw, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->Hn, w.snapH, fh, hipMemcpyDeviceToDevice, c->stream));
    const long long tw = (long long)c->KP * c->mA, th = (long long)c->KP * c->nA;
    hipLaunchKernelGGL(expand_factor_kernel, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->Wn,
                       c->WT, c->WG, c->KP, c->KP, (long long)c->mA);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(expand_factor_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->Hn,
                       c->HT, c->HG, c->KP, c->KP, (long long)c->nA);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return NBMF_OK;
  };
  if (res[2] != 0 || abort_word != 0 || res[0] < 1 || res[0] > max_iter) {
    // a barrier was abandoned (workgroups not co-resident for too long): back to the state at entry, and this
    // context keeps to the five-kernel path from now on
    if (getenv("NBMF_DEBUG")) fprintf(stderr, "[nbmf] persistent fit abandoned (status %d, abort word %llu): five-kernel path\n", res[2], abort_word);
    return back_to_entry(/*disable=*/true);
  }
  if (res[1] == 0) {   // the final factors sit in the second set of images
    HIPCHK(hipMemcpyAsync(c->Wn, w.Wn, fw, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->WT, w.WT, fw, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->WG, w.WG, fw, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->Hn, w.Hn, fh, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->HT, w.HT, fh, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->HG, w.HG, fh, hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(hipMemcpyAsync(losses, c->losses_d, sizeof(double) * (size_t)res[0], hipMemcpyDeviceToHost, c->stream));
  HIPCHK(stream_wait_spin(c->stream));
  if (small_guard_on()) {   // (see small_guard: the last loss as reported against the loss of the factors as they are in memory)
    bool ok = true;
    if (int rc = small_guard(c, losses[res[0] - 1], &ok)) return rc;
    if (!ok) return back_to_entry(/*disable=*/false);   // *handled stays false: the launches redo the run from the state at entry
  }
  *n_iter = res[0];
  *handled = true;
  g_engine_persistent_served.fetch_add(1, std::memory_order_relaxed);
  if (c->progress && c->progress_every > 0)   // the reports a launch-by-launch run would have made, in the same batches
    for (int first = 0; first < res[0]; first += c->progress_every)
      c->progress(c->progress_user, first, std::min(c->progress_every, res[0] - first), losses + first);
  return NBMF_OK;
}

// ---- several independent fits of the context's data in ONE persistent launch (nbmf_run_batch) -----------------------
// Problem p gets its own two parities of factor images, hand-off words, partial buffers, loss curve and result;
// the grid is (G workgroups) x (B problems), every workgroup on a CU of its own, and the workgroups of one problem
// synchronise among themselves only.  B = as many problems as the chip holds at once; more run as further launches.
// The arithmetic of a problem is exactly that of the single-problem launch (same kernel code, same G, PH, PW): a
// batched fit is bitwise the fit nbmf_run would have produced.
struct SmallBatchLayout {
  size_t fw, fh, factors, ll_b, pr_b, pb_b, pl_b, sync_b, res_b, pf_b, control, off_partials, off_control, off_losses, stride;
};
SmallBatchLayout small_batch_layout(const nbmf_ctx* c, int losses_cap) {
  SmallBatchLayout L{};
  const size_t g_max = 16 * SM_TPW;
  L.fw = (size_t)c->KP * c->mA * sizeof(double);
  L.fh = (size_t)c->KP * c->nA * sizeof(double);
  L.factors = 2 * (3 * L.fw + 3 * L.fh);                          // two parities of [Wn WT WG Hn HT HG]
  L.ll_b = round_up(sizeof(double) * 2 * g_max, 256);
  L.pr_b = sizeof(double) * 3 * g_max * 8 * 2;
  L.pb_b = sizeof(double) * SMALL_MAX_WGS * 2 * 16 * (size_t)c->KP;
  L.pl_b = round_up(sizeof(double) * SMALL_MAX_WGS, 256);
  L.sync_b = round_up(sizeof(unsigned long long) * (SMALL_MAX_WGS + 8), 256);
  L.res_b = 256;
  L.pf_b = round_up(sizeof(unsigned long long) * SMALL_MAX_WGS, 256);
  L.control = L.sync_b + L.res_b + L.pf_b;                        // cleared before every launch
  L.off_partials = L.factors;
  L.off_control = L.off_partials + L.ll_b + L.pr_b + L.pb_b + L.pl_b;
  L.off_losses = L.off_control + L.control;
  L.stride = round_up(L.off_losses + sizeof(double) * (size_t)losses_cap, 256);
  return L;
}

int small_batch_prepare(nbmf_ctx* c, int cap, int max_iter) {
  auto& b = c->small_batch;
  if (cap > b.cap || max_iter > b.losses_cap) {
    cap = std::max(cap, b.cap);
    const int lcap = std::max(max_iter, b.losses_cap);
    for (void* p : {(void*)b.slab, (void*)b.table, (void*)b.io})
      if (p) HIPCHK(dfree(p));
    b.slab = nullptr;
    b.table = nullptr;
    b.io = nullptr;
    b.cap = b.losses_cap = 0;
    const SmallBatchLayout L = small_batch_layout(c, lcap);
    const size_t total = (size_t)cap * L.stride;
    HIPCHK(dmalloc(&b.slab, total));
    HIPCHK(hipMemsetAsync(b.slab, 0, total, c->stream));   // pad strips of the images are never written with anything but zero
    HIPCHK(dmalloc(&b.table, sizeof(double) * 2 * (size_t)cap));   // [alpha - 1 | beta - 1] of the problems of a launch
    b.io_doubles = (size_t)cap * ((size_t)c->k * c->m + (size_t)c->k * c->n);
    HIPCHK(dmalloc(&b.io, sizeof(double) * b.io_doubles));
    b.cap = cap;
    b.losses_cap = lcap;
  }
  return NBMF_OK;
}

// *handled = false: the problems do not qualify for the persistent kernel (or a launch gave up): the caller runs them
// one by one.  Nothing of the context's own factor state is touched here.
int run_small_batch(nbmf_ctx* c, int nprob, const double* alpha, const double* beta, const double* W0, const double* H0,
                    int max_iter, double tol, double* losses, int* n_iter, double* W_out, double* H_out, bool* handled) {
  *handled = false;
  if (!small_eligible(c, c->cus)) return NBMF_OK;
  SmallArgs a{};
  small_common_args(c, max_iter, tol, &a);
  const int cap_cus = std::min(c->cus, SMALL_MAX_WGS);
  int B = std::max(1, cap_cus / a.G);
  if (const char* e = getenv("NBMF_BATCH_MAX")) B = std::max(1, std::min(B, atoi(e)));
  B = std::min(B, nprob);
  if (int rc = small_batch_prepare(c, B, max_iter)) return rc;
  auto& b = c->small_batch;
  const SmallBatchLayout L = small_batch_layout(c, b.losses_cap);
  const bool split = a.PH > 1 || a.PW > 1;
  const void* f = small_kernel_for(c, split, true);
  if (!f) return NBMF_OK;
  const int NW = sm_waves(c->KB);
  const size_t lds_bytes = sizeof(double) * ((size_t)NW * 2 * c->KB * 4 * 64 + NW * 16 + 64 + 2 * c->KP * 16) + LOG_TABLE_BYTES;
  HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  const size_t wsz = (size_t)c->k * c->m, hsz = (size_t)c->k * c->n;
  // problem 0's pointers; problem p's are these plus p * stride (PP() in the kernel, prob() here)
  {
    char* q = b.slab;
    for (int par = 0; par < 2; ++par) {
      a.Wn[par] = (double*)q; q += L.fw;
      a.WT[par] = (double*)q; q += L.fw;
      a.WG[par] = (double*)q; q += L.fw;
      a.Hn[par] = (double*)q; q += L.fh;
      a.HT[par] = (double*)q; q += L.fh;
      a.HG[par] = (double*)q; q += L.fh;
    }
    a.ll_part = (double*)q; q += L.ll_b;
    a.prior_part = (double*)q; q += L.pr_b;
    a.part_buf = (double*)q; q += L.pb_b;
    a.part_ll = (double*)q;
    char* ctl = b.slab + L.off_control;
    a.sync = (unsigned long long*)ctl;
    a.result = (int*)(ctl + L.sync_b);
    a.part_flag = (unsigned long long*)(ctl + L.sync_b + L.res_b);
    a.losses = (double*)(b.slab + L.off_losses);
    a.prob_stride = L.stride;
    a.am1s = (const double*)b.table;
    a.bm1s = (const double*)b.table + b.cap;
  }
  auto prob = [&](auto* p0, int p) { return (decltype(p0))((char*)p0 + (size_t)p * L.stride); };
  std::vector<int> res((size_t)B * 4);
  std::vector<unsigned long long> abort_words((size_t)B);
  std::vector<double> priors((size_t)2 * b.cap);
  for (int p0 = 0; p0 < nprob; p0 += B) {
    const int Bc = std::min(B, nprob - p0);
    for (int p = 0; p < Bc; ++p) {
      priors[(size_t)p] = alpha[p0 + p] - 1.0;
      priors[(size_t)b.cap + p] = beta[p0 + p] - 1.0;
    }
    HIPCHK(hipMemcpyAsync(b.table, priors.data(), sizeof(double) * priors.size(), hipMemcpyHostToDevice, c->stream));
    // initial factors of the whole chunk in one copy each, expanded into every problem's parity-1 images
    HIPCHK(hipMemcpyAsync(b.io, W0 + (size_t)p0 * wsz, sizeof(double) * wsz * Bc, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(b.io + wsz * Bc, H0 + (size_t)p0 * hsz, sizeof(double) * hsz * Bc, hipMemcpyHostToDevice, c->stream));
    for (int p = 0; p < Bc; ++p) {
      const long long tw = (long long)c->KP * c->mA, th = (long long)c->KP * c->nA;
      hipLaunchKernelGGL(set_factor_kernel, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0, c->stream, (const double*)(b.io + wsz * p),
                         prob(a.Wn[1], p), prob(a.WT[1], p), prob(a.WG[1], p), c->k, c->KP, c->KP, (long long)c->m, (long long)c->mA);
      hipLaunchKernelGGL(set_factor_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, c->stream,
                         (const double*)(b.io + wsz * Bc + hsz * p), prob(a.Hn[1], p), prob(a.HT[1], p), prob(a.HG[1], p), c->k, c->KP,
                         c->KP, (long long)c->n, (long long)c->nA);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemset2DAsync(b.slab + L.off_control, L.stride, 0, L.control, (size_t)Bc, c->stream));   // every problem's control words
    if (getenv("NBMF_SMALL_FORCE_ABORT")) {   // tests: the abort word of the first problem up front
      const unsigned long long one = 1;
      HIPCHK(hipMemcpyAsync(a.sync + a.G, &one, sizeof one, hipMemcpyHostToDevice, c->stream));
    }
    void* params[] = {&a};
    {
      SmallReservation cus_held(c->device, a.G * Bc, cap_cus);
      struct DrainGuard {
        hipStream_t st;
        ~DrainGuard() { hipStreamSynchronize(st); }
      } drain_guard{c->stream};
      HIPCHK(hipLaunchKernel(f, dim3(a.G, Bc), dim3(64 * NW), params, lds_bytes, c->stream));
      // results and abort words of all problems: strided copies
      HIPCHK(hipMemcpy2DAsync(res.data(), sizeof(int) * 4, a.result, L.stride, sizeof(int) * 4, (size_t)Bc, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipMemcpy2DAsync(abort_words.data(), sizeof(unsigned long long), a.sync + a.G, L.stride, sizeof(unsigned long long), (size_t)Bc,
                              hipMemcpyDeviceToHost, c->stream));
      HIPCHK(stream_wait_spin(c->stream));
    }
    ++b.launches;
    b.problems += Bc;
    for (int p = 0; p < Bc; ++p)
      if (res[(size_t)p * 4 + 2] != 0 || abort_words[(size_t)p] != 0 || res[(size_t)p * 4] < 1 || res[(size_t)p * 4] > max_iter) {
        c->small.disabled = true;   // a barrier was abandoned: this context keeps to the launches from now on
        ++c->small.aborted;
        g_engine_persistent_aborted.fetch_add(1, std::memory_order_relaxed);
        if (getenv("NBMF_DEBUG")) fprintf(stderr, "[nbmf] batched persistent fit abandoned (problem %d): one by one\n", p0 + p);
        return NBMF_OK;             // *handled stays false: the caller redoes ALL problems one by one
      }
    // the end-of-run guard (small_guard), on ONE problem of every launch, taken in turn: its last reported loss against the
    // loss of its final factors recomputed by the launches -- on the problem's own images (the context's pointers are lent
    // to them for the three launches: no copy, and the context's own factors are untouched)
    if (small_guard_on()) {
      const int ps = (int)(b.launches % Bc), par = res[(size_t)ps * 4 + 1], nit = res[(size_t)ps * 4];
      double reported = 0.0;
      HIPCHK(hipMemcpyAsync(&reported, (const char*)a.losses + (size_t)ps * L.stride + sizeof(double) * (size_t)(nit - 1), sizeof(double),
                            hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      struct Lend {   // (restored on every path out of this block)
        nbmf_ctx* c;
        double *Wn, *WT, *WG, *Hn, *HT, *HG, alpha, beta;
        ~Lend() {
          c->Wn = Wn; c->WT = WT; c->WG = WG; c->Hn = Hn; c->HT = HT; c->HG = HG;
          c->alpha = alpha; c->beta = beta;
        }
      } lend{c, c->Wn, c->WT, c->WG, c->Hn, c->HT, c->HG, c->alpha, c->beta};
      c->Wn = prob(a.Wn[par], ps); c->WT = prob(a.WT[par], ps); c->WG = prob(a.WG[par], ps);
      c->Hn = prob(a.Hn[par], ps); c->HT = prob(a.HT[par], ps); c->HG = prob(a.HG[par], ps);
      c->alpha = alpha[p0 + ps];
      c->beta = beta[p0 + ps];
      bool ok = true;
      if (int rc = small_guard(c, reported, &ok)) return rc;
      if (!ok) {
        ++c->small.aborted;
        g_engine_persistent_aborted.fetch_add(1, std::memory_order_relaxed);
        return NBMF_OK;             // *handled stays false: the caller redoes ALL problems one by one, by the launches
      }
    }
    // results: final factors (the parity each problem ended on) through one staging copy, the loss curves in one strided copy
    for (int p = 0; p < Bc; ++p) {
      const int par = res[(size_t)p * 4 + 1];
      const long long tw = (long long)c->k * c->m, th = (long long)c->k * c->n;
      hipLaunchKernelGGL(get_factor_kernel, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0, c->stream, (const double*)prob(a.Wn[par], p),
                         b.io + wsz * p, c->k, (long long)c->m, (long long)c->mA);
      hipLaunchKernelGGL(get_factor_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, c->stream, (const double*)prob(a.Hn[par], p),
                         b.io + wsz * Bc + hsz * p, c->k, (long long)c->n, (long long)c->nA);
      n_iter[p0 + p] = res[(size_t)p * 4];
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy2DAsync(losses + (size_t)p0 * max_iter, sizeof(double) * (size_t)max_iter, a.losses, L.stride, sizeof(double) * (size_t)max_iter,
                            (size_t)Bc, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(W_out + (size_t)p0 * wsz, b.io, sizeof(double) * wsz * Bc, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(H_out + (size_t)p0 * hsz, b.io + wsz * Bc, sizeof(double) * hsz * Bc, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait_spin(c->stream));
  }
  c->small.runs += nprob;
  g_engine_persistent_served.fetch_add(nprob, std::memory_order_relaxed);
  *handled = true;
  return NBMF_OK;
}

int ensure_losses(nbmf_ctx* c, int cap) {
  if (cap <= c->losses_cap) return NBMF_OK;
  if (c->losses_d) HIPCHK(dfree(c->losses_d));
  c->losses_d = nullptr;
  HIPCHK(dmalloc(&c->losses_d, sizeof(double) * (size_t)cap));
  c->losses_cap = cap;
  return NBMF_OK;
}

// After an upload of binary data that is observed everywhere: see mask_pad_rows_kernel (the two-state W sweep then
// applies whatever the shape).  Called once n_obs is known; setup_workspaces has just rebuilt the lane masks.
int mark_pad_rows_b(nbmf_ctx* c) {
  c->padB_ones = false;
  if (c->data_kind != DATA_BIN || !c->bitsB || c->n_obs != (double)c->m * (double)c->n || c->nA == c->n) return NBMF_OK;
  const long long RbB = c->nA / 16, CbB = c->mA / 16, nblk = RbB - (c->n >> 4);
  const long long threads = CbB * nblk * 4;
  hipLaunchKernelGGL(mask_pad_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, c->bitsB, RbB, CbB,
                     (long long)c->n);
  HIPCHK(hipGetLastError());
  c->padB_ones = true;
  return NBMF_OK;
}

// Chunking of the two sweeps and the slabs that go with it; needs the storage path (it decides the
// kernels' residency), so it runs at the end of nbmf_upload.
int setup_workspaces(nbmf_ctx* c) {
  const int cus = c->cus;
  const int NB = 8 / c->KB;
  const int slotsH = cus * resident_per_cu<MODE_H>(c->KB, c->data_kind);
  const int slotsW = cus * resident_per_cu<MODE_W>(c->KB, c->data_kind);
  const int ns = wg_strips(c) / WG_WAVES;
  pick_chunks((int)(c->nA / 16 / wg_strips(c)), (int)(c->mA / 16), NB, slotsH, ns, cus, /*one_round_ok=*/true, &c->chunksH, &c->CH_H);
  pick_chunks((int)(c->mA / 16 / wg_strips(c)), (int)(c->nA / 16), NB, slotsW, ns, cus, /*one_round_ok=*/false, &c->chunksW, &c->CH_W);
  const std::vector<int> bH = chunk_boundaries((int)(c->mA / 16), c->CH_H);
  const std::vector<int> bW = chunk_boundaries((int)(c->nA / 16), c->CH_W);
  c->chunksH = (int)bH.size() - 1;
  c->chunksW = (int)bW.size() - 1;
  c->bW_host = bW;
  for (int** p : {&c->cstartH, &c->cstartW}) {
    if (*p) HIPCHK(dfree(*p));
    *p = nullptr;
  }
  HIPCHK(dmalloc(&c->cstartH, sizeof(int) * bH.size()));
  HIPCHK(dmalloc(&c->cstartW, sizeof(int) * bW.size()));
  HIPCHK(hipMemcpy(c->cstartH, bH.data(), sizeof(int) * bH.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->cstartW, bW.data(), sizeof(int) * bW.size(), hipMemcpyHostToDevice));
  if (getenv("NBMF_DEBUG"))
    fprintf(stderr, "[nbmf] K_pad=%d path=%d: H-pass %d x %d workgroups (chunk %d blocks, %d slots), W-pass %d x %d (chunk %d, %d slots)\n",
            c->KP, c->data_kind, (int)(c->nA / 64), c->chunksH, c->CH_H, slotsH, (int)(c->mA / 64), c->chunksW, c->CH_W, slotsW);
  for (double** p : {&c->slabH, &c->slabW, &c->lossbuf, &c->lossfin}) {
    if (*p) HIPCHK(dfree(*p));
    *p = nullptr;
  }
  // binary path: the lane-mask images of the two code images (what the sweeps read)
  for (unsigned long long** p : {&c->bitsA, &c->bitsB}) {
    if (*p) HIPCHK(dfree(*p));
    *p = nullptr;
  }
  if (c->data_kind == DATA_BIN) {
    const long long n_tiles = (long long)(c->mA / 16) * (c->nA / 16);
    HIPCHK(dmalloc(&c->bitsA, sizeof(unsigned long long) * 8 * (size_t)n_tiles + PASS_SLACK));   // (slack: record requests and
    HIPCHK(dmalloc(&c->bitsB, sizeof(unsigned long long) * 8 * (size_t)n_tiles + PASS_SLACK));   //  prefetches run past the end)
    for (int image = 0; image < 2; ++image) {
      hipLaunchKernelGGL(mask_build_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, c->stream,
                         (const uint32_t*)(image == 0 ? c->dataA : c->dataB), image == 0 ? c->bitsA : c->bitsB, n_tiles);
      HIPCHK(hipGetLastError());
    }
  }
  const size_t fw = (size_t)c->KP * c->mA * sizeof(double), fh = (size_t)c->KP * c->nA * sizeof(double);
  HIPCHK(dmalloc(&c->slabH, 2 * (size_t)c->chunksH * fh));
  HIPCHK(dmalloc(&c->slabW, (size_t)c->chunksW * fw));
  const size_t n_slots = (size_t)c->chunksH * (c->nA / 16 / WG_WAVES);
  HIPCHK(dmalloc(&c->lossbuf, sizeof(double) * n_slots));
  HIPCHK(dmalloc(&c->lossfin, sizeof(double) * n_slots));
  if (int rc = fill_ll_empty(c)) return rc;
  return NBMF_OK;
}

// Drop whatever communicator is attached (the stream must be idle); the context is single-GPU again.
void comm_release(nbmf_ctx* c, bool recount = false) {
  // the column split replaced the per-row observed counts by their global sums: back to this shard's own
  if (recount && is_sharded(c) && c->shard_axis == 1 && c->data_kind >= 0) {
    hipLaunchKernelGGL(rowcount_kernel, dim3((unsigned)((c->m + 255) / 256)), dim3(256), 0, c->stream, c->dataB, c->maskB,
                       c->data_kind, (long long)(c->nA / 16), (long long)c->m, (long long)c->n, c->rowcnt);
    hipStreamSynchronize(c->stream);
  }
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  c->comm = nullptr;
  c->host_reduce = nullptr;
  c->host_reduce_user = nullptr;
  if (c->peer) {
    c->Pbuf = c->Pbuf_own;
    c->sbuf = c->sbuf_own;
    c->Qbuf = c->Qbuf_own;
    c->Pbuf_own = c->sbuf_own = c->Qbuf_own = nullptr;
  }
  for (void* q : c->peer_mapped) hipIpcCloseMemHandle(q);
  c->peer_mapped.clear();
  c->peer = false;
  c->prior_src = c->prior;
  c->n_prior_src = c->n_prior_blocks;
  c->ll_ptr = nullptr;
  c->nranks = 1;
  c->rank = 0;
  c->shard_axis = 0;
  c->npanel = 1;
  c->n_obs_global = c->n_obs;
  c->n_div_global = (double)c->n;
}

// After a stream synchronisation: did a peer exchange give up waiting?
int peer_check(nbmf_ctx* c) {
  if (!c->peer) return NBMF_OK;
  unsigned long long err = 0;
  HIPCHK(hipMemcpy(&err, c->pflags + PF_ERR, sizeof err, hipMemcpyDeviceToHost));
  if (err)
    return fail(NBMF_ERR_COMM, "peer exchange %llu timed out on rank %d (a rank died, stalled for more than "
                               "NBMF_PEER_TIMEOUT_MS, or ran a different sequence of calls); detach the communicator", err, c->rank);
  return NBMF_OK;
}

int ready(nbmf_ctx* c) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (c->data_kind < 0) return fail(NBMF_ERR_STATE, "nbmf_upload has not been called");
  if (!c->have_factors) return fail(NBMF_ERR_STATE, "nbmf_set_factors has not been called");
  return NBMF_OK;
}

int set_device(nbmf_ctx* c) {
  HIPCHK(hipSetDevice(c->device));
  return NBMF_OK;
}

// (also the single-launch engine's end-of-run guard: small_guard)
int loss_by_launches(nbmf_ctx* c, double* loss) {
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_losses(c, 1)) return rc;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));
  hipLaunchKernelGGL(prior_kernel, dim3(c->n_prior_blocks), dim3(256), 0, c->stream, c->Hn, c->prior, c->k, c->KP,
                     (long long)c->n, (long long)c->nA, c->eps);
  HIPCHK(hipGetLastError());
  c->prior_src = c->prior;
  c->n_prior_src = c->n_prior_blocks;
  if (int rc = enqueue_loglik_pass(c, 0)) return rc;
  if (int rc = enqueue_finalize(c, 0, 0.0)) return rc;
  HIPCHK(hipMemcpyAsync(loss, c->losses_d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->timing) timing_collect(c);
  return peer_check(c);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int nbmf_abi_version(void) { return 4; }   // 2: NBMF_PEER_HANDLE_BYTES 128 -> 192, nbmf_generate_slice, nbmf_set_storage; 3: nbmf_upload_v (uint8 / bool data), nbmf_selftest_mfma_peak, nbmf_engine_stats; 4: nbmf_source_hash, nbmf_comm_info, nbmf_cancel, nbmf_variant_stats, nbmf_device_bus_id
// The sources this binary was compiled from: first 12 hex digits of the SHA-256 over nbmf_hip.hip, the *.inc files (sorted
// by name) and include/nbmf_hip.h, put in by the Makefile (tools/src_hash.sh prints the same for the tree).  A profile or a
// bench line that quotes it can be tied to a commit; a library built by hand without the Makefile says "unstamped".
#ifndef NBMF_SRC_HASH
#define NBMF_SRC_HASH "unstamped"
#endif
const char* nbmf_source_hash(void) { return NBMF_SRC_HASH; }

const char* nbmf_last_error(void) { return g_err.c_str(); }

int nbmf_device_bus_id(int device, char* buf, int len) {
  if (!buf || len < 16) return fail(NBMF_ERR_ARG, "buffer of at least 16 bytes expected");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(NBMF_ERR_HIP, "no HIP device %d", device);
  HIPCHK(hipDeviceGetPCIBusId(buf, len, device));
  return NBMF_OK;
}

int nbmf_device_count(int* count) {
  if (!count) return fail(NBMF_ERR_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(NBMF_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return NBMF_OK;
}

int nbmf_create(int64_t m, int64_t n, int k, int device, nbmf_ctx** out) {
  if (!out) return fail(NBMF_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (m < 1 || n < 1) return fail(NBMF_ERR_ARG, "m and n must be >= 1 (got %lld x %lld)", (long long)m, (long long)n);
  if (k < 1 || k > NBMF_MAX_K) return fail(NBMF_ERR_ARG, "n_components must be in [1, %d] (got %d)", NBMF_MAX_K, k);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return fail(NBMF_ERR_HIP, "no HIP device available (%s): libnbmf_hip needs an MI355X (gfx950)",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(NBMF_ERR_ARG, "device %d out of range (have %d)", device, ndev);
  HIPCHK(hipSetDevice(device));
  DevInfo info;
  HIPCHK(device_info(device, &info));
  if (!info.gfx950) return fail(NBMF_ERR_HIP, "device %d is %s; this library is built for gfx950 only", device, info.arch);

  nbmf_ctx* c = new nbmf_ctx();
  c->cus = info.cus;
  struct CtxGuard {   // a failure below releases what has been allocated so far
    nbmf_ctx* p;
    ~CtxGuard() {
      if (p) nbmf_destroy(p);
    }
  } ctx_guard{c};
  c->device = device;
  c->m = m;
  c->n = n;
  c->k = k;
  c->mA = round_up(m, PAD);
  c->nA = round_up(n, PAD);
  c->KB = k <= 16 ? 1 : k <= 32 ? 2 : k <= 64 ? 4 : 8;
  c->KS = (k + SLICE_K - 1) / SLICE_K;
  c->KP = c->KS > 1 ? SLICE_K * c->KS : 16 * c->KB;
  if (c->KS > 1) HIPCHK(dmalloc(&c->theta, sizeof(double) * (size_t)c->mA * c->nA));
  HIPCHK(stream_acquire(device, &c->stream));
  HIPCHK(log_table_build(device, c->stream));   // (the general path's table: once per device, never in the launch path)

  const size_t fw = (size_t)c->KP * c->mA * sizeof(double), fh = (size_t)c->KP * c->nA * sizeof(double);
  HIPCHK(dmalloc(&c->Wn, fw));
  HIPCHK(dmalloc(&c->WT, fw + PASS_SLACK));   // (+ slack: the sweeps prefetch one stage past the end, see STAGE_DMA)
  HIPCHK(dmalloc(&c->WG, fw + PASS_SLACK));
  HIPCHK(dmalloc(&c->Hn, fh));
  HIPCHK(dmalloc(&c->HT, fh + PASS_SLACK));
  HIPCHK(dmalloc(&c->HG, fh + PASS_SLACK));
  // (the pass workspaces depend on the storage path: setup_workspaces, called by nbmf_upload)
  HIPCHK(dmalloc(&c->Pbuf, 2 * fh + 64));
  c->n_prior_blocks = (int)(((size_t)c->KP * c->nA + 255) / 256);
  HIPCHK(dmalloc(&c->prior, sizeof(double) * 2 * (size_t)c->n_prior_blocks));
  c->prior_src = c->prior;
  c->n_prior_src = c->n_prior_blocks;
  HIPCHK(dmalloc(&c->scal, sizeof(double) * 8));
  HIPCHK(dmalloc(&c->sbuf, sizeof(double) * 8));
  HIPCHK(dmalloc(&c->flags, sizeof(int) * 8));
  HIPCHK(dmalloc(&c->stats, sizeof(unsigned long long) * 8));
  HIPCHK(dmalloc(&c->rowcnt, sizeof(double) * (size_t)c->mA));
  HIPCHK(hipMemset(c->flags, 0, sizeof(int) * 8));
  HIPCHK(hipMemset(c->scal, 0, sizeof(double) * 8));
  c->stage_bytes = (fw > fh ? fw : fh);
  HIPCHK(dmalloc(&c->stage, c->stage_bytes));
  ctx_guard.p = nullptr;
  *out = c;
  return NBMF_OK;
}

int nbmf_destroy(nbmf_ctx* c) {
  if (!c) return NBMF_OK;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  comm_release(c);
  if (c->host_buf) hipHostFree(c->host_buf);
  arena_release((ArenaSlot*)c->arena_slot.p);   // back to the pool, never to the allocator (see ArenaSlot)
  void* ptrs[] = {c->dataA, c->dataB, c->maskA, c->maskB, c->rowcnt, c->Wn, c->WT, c->WG, c->Hn, c->HT, c->HG,
                  c->slabH, c->slabW, c->Pbuf, c->lossbuf, c->lossfin, c->prior, c->scal, c->flags, c->losses_d, c->stage, c->stats,
                  c->sbuf, c->Qbuf, c->cstartH, c->cstartW, c->theta, c->small.slab, c->small_batch.slab, c->small_batch.table,
                  c->small_batch.io, c->bitsA, c->bitsB};
  for (void* p : ptrs)
    if (p) dfree(p);
  for (hipEvent_t e : c->ev) hipEventDestroy(e);
  for (hipEvent_t e : {c->evH, c->evF, c->ev1})
    if (e) hipEventDestroy(e);
  if (c->stream2) {
    hipStreamSynchronize(c->stream2);
    hipStreamDestroy(c->stream2);
  }
  if (c->stream) stream_release(c->device, c->stream);   // (synchronised at the top of this function)
  delete c;
  return NBMF_OK;
}

int nbmf_set_hyper(nbmf_ctx* c, double alpha, double beta, double eps, int projection) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (projection != NBMF_PROJ_NORMALIZE && projection != NBMF_PROJ_DUCHI)
    return fail(NBMF_ERR_ARG, "unknown projection %d", projection);
  if (!(eps >= 2.2250738585072014e-308 && eps <= 1.7976931348623157e308))
    return fail(NBMF_ERR_ARG, "eps must be a positive normal number");
  c->alpha = alpha;
  c->beta = beta;
  c->eps = eps;
  c->projection = projection;
  return NBMF_OK;
}

int nbmf_upload(nbmf_ctx* c, const double* x, int64_t ldx, int transposed, const void* mask, int mask_kind,
                int64_t ldmask, int* out_flags) {
  return nbmf_upload_v(c, x, NBMF_DATA_F64, ldx, transposed, mask, mask_kind, ldmask, out_flags);
}

int nbmf_upload_v(nbmf_ctx* c, const void* xv, int x_kind, int64_t ldx, int transposed, const void* mask, int mask_kind,
                  int64_t ldmask, int* out_flags) {
  if (!c || !xv) return fail(NBMF_ERR_ARG, "null context or data");
  if (x_kind != NBMF_DATA_F64 && x_kind != NBMF_DATA_U8 && x_kind != NBMF_DATA_F32)
    return fail(NBMF_ERR_ARG, "x_kind must be NBMF_DATA_F64, NBMF_DATA_U8 or NBMF_DATA_F32");
  const size_t xsz = x_kind == NBMF_DATA_U8 ? 1 : (x_kind == NBMF_DATA_F32 ? 4 : 8);   // bytes per element of the host array
  const double* x = (const double*)xv;
  const unsigned char* xb = (const unsigned char*)xv;
  if (mask_kind != NBMF_MASK_NONE && !mask) return fail(NBMF_ERR_ARG, "mask_kind set but mask is NULL");
  if (!mask) mask_kind = NBMF_MASK_NONE;
  if (is_sharded(c)) return fail(NBMF_ERR_STATE, "upload before attaching a communicator");
  if (int rc = set_device(c)) return rc;
  const int64_t U = transposed ? c->n : c->m, V = transposed ? c->m : c->n;
  if (ldx < V || (mask && ldmask < V)) return fail(NBMF_ERR_ARG, "leading dimension smaller than the row length");

  // cheap host-side guess of the storage path from a sample (the device pack verifies it exactly);
  // NBMF_FORCE_F64=1 keeps binary data on the 8-byte path (measurement only)
  bool guess_bin = c->storage == NBMF_STORAGE_AUTO && !(getenv("NBMF_FORCE_F64") && atoi(getenv("NBMF_FORCE_F64")) != 0);
  bool guess_mask_bin = true;   // a float64 mask holding only 0 and 1 is a binary mask
  {
    const int64_t rows = U < 8 ? U : 8;
    for (int64_t u = 0; u < rows && (guess_bin || guess_mask_bin); ++u) {
      const int64_t uu = (U - 1) * u / (rows > 1 ? rows - 1 : 1);
      for (int64_t v = 0; v < V && v < 4096; ++v) {
        const double xe = x_kind == NBMF_DATA_U8 ? (double)xb[uu * ldx + v]
                          : x_kind == NBMF_DATA_F32 ? (double)((const float*)xv)[uu * ldx + v] : x[uu * ldx + v];
        if (xe != 0.0 && xe != 1.0) guess_bin = false;
        if (mask_kind == NBMF_MASK_F64) {
          const double mk = ((const double*)mask)[uu * ldmask + v];
          if (mk != 0.0 && mk != 1.0) guess_mask_bin = false;
        }
      }
    }
    if (!guess_mask_bin) guess_bin = false;
  }

  // NBMF_UPLOAD_TRACE=1: where an upload's time goes, stage by stage (host clock, the stream drained at every mark)
  const bool up_trace = getenv("NBMF_UPLOAD_TRACE") != nullptr;
  auto up_t0 = std::chrono::steady_clock::now();
  double up_copy = 0, up_pack = 0;
  auto up_mark = [&](const char* what) {
    if (!up_trace) return;
    hipStreamSynchronize(c->stream);
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[nbmf upload] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - up_t0).count());
    up_t0 = t;
  };
  up_mark("host sample");
  const size_t tiles = (size_t)(c->mA / 16) * (c->nA / 16);
  // staging chunk: <= 256 MiB of raw rows and <= 32768 tile rows (grid.y limit of the pack launch)
  const int64_t chunk_rows_max =
      std::min<int64_t>(32768 * 16, std::max<int64_t>(PAD, ((int64_t)(256ll << 20) / (V * (int64_t)xsz)) / PAD * PAD));
  char* raw = nullptr;
  void* rawm = nullptr;
  struct StagingGuard {   // the raw staging buffers never outlive this call, whichever way it returns
    char*& a;
    void*& b;
    ~StagingGuard() {
      if (a) dfree(a);
      if (b) dfree(b);
    }
  } staging_guard{raw, rawm};
  const size_t msz = mask_kind == NBMF_MASK_F64 ? 8 : 1;
  const int64_t chunk_rows = std::min<int64_t>(round_up(U, PAD), chunk_rows_max);
  HIPCHK(dmalloc(&raw, (size_t)chunk_rows * V * xsz));
  if (mask) HIPCHK(dmalloc(&rawm, (size_t)chunk_rows * V * msz));

  up_mark("staging buffers");
  int rc = NBMF_OK;
  unsigned long long st[4] = {0, 0, 0, 0};
  // Storage paths, cheapest first: byte codes (binary data, binary or no mask); doubles with a binary (or no)
  // mask folded in as NaN; doubles plus weight tiles.  A path the sample suggested is dropped when the pack's
  // exact counts say otherwise, and the next one is packed.
  for (int kind : {DATA_BIN, DATA_F64, DATA_F64M}) {
    if (kind == DATA_BIN && !guess_bin) continue;
    if (kind == DATA_F64 && (!guess_mask_bin || c->storage == NBMF_STORAGE_F64_WEIGHTS)) continue;
    if (kind == DATA_F64M && mask_kind == NBMF_MASK_NONE && c->storage != NBMF_STORAGE_F64_WEIGHTS) continue;   // (no mask: all-ones weights)
    const bool binary = kind == DATA_BIN;
    for (void** p : {&c->dataA, &c->dataB, &c->maskA, &c->maskB}) {
      if (*p) dfree(*p);
      *p = nullptr;
    }
    c->data_kind = -1;
    const size_t esz = binary ? 1 : 8;
    const size_t bytes = tiles * 256 * esz;
    HIPCHK(dmalloc(&c->dataA, bytes + PASS_SLACK));
    HIPCHK(dmalloc(&c->dataB, bytes + PASS_SLACK));
    // (the pack kernel writes every tile of the padded mA x nA grid, pad entries included)
    if (kind == DATA_F64M) {
      HIPCHK(dmalloc(&c->maskA, bytes + PASS_SLACK));   // (slack: the sweeps request the tile behind their last one)
      HIPCHK(dmalloc(&c->maskB, bytes + PASS_SLACK));
    }
    HIPCHK(hipMemsetAsync(c->stats, 0, sizeof(unsigned long long) * 8, c->stream));
    up_mark("image allocations");

    for (int64_t u0 = 0; u0 < round_up(U, PAD); u0 += chunk_rows) {
      const int64_t urows_pad = std::min<int64_t>(chunk_rows, round_up(U, PAD) - u0);
      const int64_t urows = std::max<int64_t>(0, std::min<int64_t>(U - u0, urows_pad));
      if (urows > 0) {
        HIPCHK(hipMemcpy2DAsync(raw, (size_t)V * xsz, xb + (size_t)u0 * ldx * xsz, (size_t)ldx * xsz, (size_t)V * xsz, (size_t)urows,
                                hipMemcpyHostToDevice, c->stream));
        if (mask)
          HIPCHK(hipMemcpy2DAsync(rawm, (size_t)V * msz, (const char*)mask + (size_t)u0 * ldmask * msz,
                                  (size_t)ldmask * msz, (size_t)V * msz, (size_t)urows, hipMemcpyHostToDevice,
                                  c->stream));
      }
      if (up_trace) {
        hipStreamSynchronize(c->stream);
        const auto t = std::chrono::steady_clock::now();
        up_copy += std::chrono::duration<double, std::milli>(t - up_t0).count();
        up_t0 = t;
      }
      PackArgs a{};
      a.x = raw;
      a.x_kind = x_kind;
      a.mask = mask ? rawm : nullptr;
      a.mask_kind = mask_kind;
      a.ldx = V;
      a.ldmask = V;
      a.u0 = u0;
      a.urows = urows;
      a.U = U;
      a.V = V;
      a.transposed = transposed;
      a.binary = binary ? 1 : 0;
      a.fold_mask = kind == DATA_F64;
      a.dataA = c->dataA;
      a.dataB = c->dataB;
      a.maskA = c->maskA;
      a.maskB = c->maskB;
      a.RbA = c->mA / 16;
      a.RbB = c->nA / 16;
      a.stats = c->stats;
      const int64_t VA = transposed ? c->mA : c->nA;
      a.tile_rows = urows_pad / 16;
      dim3 grid((unsigned)(VA / 16 / 4), (unsigned)((a.tile_rows + PACK_TILE_ROWS - 1) / PACK_TILE_ROWS));
      hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, c->stream, a);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(c->stream));   // raw staging buffer is reused by the next chunk
      if (up_trace) {
        const auto t = std::chrono::steady_clock::now();
        up_pack += std::chrono::duration<double, std::milli>(t - up_t0).count();
        up_t0 = t;
      }
    }
    if (up_trace) fprintf(stderr, "[nbmf upload] host -> device copies %.3f ms, pack kernels %.3f ms\n", up_copy, up_pack);
    HIPCHK(hipMemcpy(st, c->stats, sizeof st, hipMemcpyDeviceToHost));
    if (st[1] != 0) {
      rc = fail(NBMF_ERR_RANGE, "X must be binary: %llu entries outside [0,1] or not finite", st[1]);
      break;
    }
    if (st[3] != 0) guess_mask_bin = false;   // exact now: weights that are not 0 or 1 need the weight tiles
    if (kind == DATA_BIN && (st[2] != 0 || st[3] != 0)) continue;   // sample guessed wrong: repack as doubles
    if (kind == DATA_F64 && st[3] != 0) continue;
    c->data_kind = kind;
    break;
  }
  if (rc != NBMF_OK) return rc;
  if (c->data_kind < 0) return fail(NBMF_ERR_STATE, "internal: pack did not settle on a storage path");
  up_mark("pack statistics");
  if (int rc2 = setup_workspaces(c)) return rc2;
  up_mark("workspaces + lane masks");
  c->n_obs = (mask_kind == NBMF_MASK_NONE) ? (double)c->m * (double)c->n : (double)st[0];
  if (int rc2 = mark_pad_rows_b(c)) return rc2;
  c->n_obs_global = c->n_obs;
  hipLaunchKernelGGL(rowcount_kernel, dim3((unsigned)((c->m + 255) / 256)), dim3(256), 0, c->stream, c->dataB, c->maskB,
                     c->data_kind, (long long)(c->nA / 16), (long long)c->m, (long long)c->n, c->rowcnt);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  up_mark("row counts");
  if (out_flags) *out_flags = (c->data_kind == DATA_BIN) ? NBMF_FLAG_BINARY_PATH : 0;
  return NBMF_OK;
}

int nbmf_generate_slice(nbmf_ctx* c, uint64_t seed, double density, double observed, int64_t row0, int64_t col0,
                        int64_t n_global) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (!(density >= 0.0 && density <= 1.0) || !(observed >= 0.0 && observed <= 1.0))
    return fail(NBMF_ERR_ARG, "density and observed must lie in [0, 1]");
  if (row0 < 0 || col0 < 0 || n_global < col0 + c->n)
    return fail(NBMF_ERR_ARG, "slice [%lld.., %lld..+%lld) does not fit a matrix %lld columns wide", (long long)row0, (long long)col0,
                (long long)c->n, (long long)n_global);
  if (is_sharded(c)) return fail(NBMF_ERR_STATE, "generate before attaching a communicator");
  if (int rc = set_device(c)) return rc;
  for (void** p : {&c->dataA, &c->dataB, &c->maskA, &c->maskB}) {
    if (*p) dfree(*p);
    *p = nullptr;
  }
  c->data_kind = -1;
  const size_t tiles = (size_t)(c->mA / 16) * (c->nA / 16);
  HIPCHK(dmalloc(&c->dataA, tiles * 256 + PASS_SLACK));
  HIPCHK(dmalloc(&c->dataB, tiles * 256 + PASS_SLACK));
  HIPCHK(hipMemsetAsync(c->stats, 0, sizeof(unsigned long long) * 8, c->stream));
  const long long RbA = c->mA / 16;
  dim3 grid((unsigned)(c->nA / 16 / 4), (unsigned)std::min<long long>(RbA, 65535), (unsigned)((RbA + 65534) / 65535));
  hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, c->stream, (uint32_t*)c->dataA, (uint32_t*)c->dataB,
                     RbA, (long long)(c->nA / 16), (long long)c->m, (long long)c->n,
                     SynthSlice{(long long)row0, (long long)col0, (long long)n_global}, (unsigned long long)seed, density, observed,
                     c->stats);
  HIPCHK(hipGetLastError());
  unsigned long long st[1] = {0};
  HIPCHK(hipMemcpyAsync(st, c->stats, sizeof st, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->data_kind = DATA_BIN;
  if (int rc2 = setup_workspaces(c)) return rc2;
  c->n_obs = (double)st[0];
  if (int rc2 = mark_pad_rows_b(c)) return rc2;
  c->n_obs_global = c->n_obs;
  hipLaunchKernelGGL(rowcount_kernel, dim3((unsigned)((c->m + 255) / 256)), dim3(256), 0, c->stream, c->dataB, c->maskB,
                     c->data_kind, (long long)(c->nA / 16), (long long)c->m, (long long)c->n, c->rowcnt);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return NBMF_OK;
}

int nbmf_generate(nbmf_ctx* c, uint64_t seed, double density, double observed) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  return nbmf_generate_slice(c, seed, density, observed, 0, 0, c->n);
}

int nbmf_set_storage(nbmf_ctx* c, int storage) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (storage < NBMF_STORAGE_AUTO || storage > NBMF_STORAGE_F64_WEIGHTS) return fail(NBMF_ERR_ARG, "unknown storage path %d", storage);
  c->storage = storage;
  return NBMF_OK;
}

int nbmf_upload_csr(nbmf_ctx* c, const int64_t* indptr, const int32_t* indices, int64_t nnz, int transposed,
                    const int64_t* mask_indptr, const int32_t* mask_indices, int64_t mask_nnz, int* out_flags) {
  if (!c || !indptr || (nnz > 0 && !indices)) return fail(NBMF_ERR_ARG, "null context or CSR arrays");
  if (nnz < 0 || mask_nnz < 0) return fail(NBMF_ERR_ARG, "negative entry count");
  const bool masked = mask_indptr != nullptr;
  if (masked && mask_nnz > 0 && !mask_indices) return fail(NBMF_ERR_ARG, "mask_indptr without mask_indices");
  if (is_sharded(c)) return fail(NBMF_ERR_STATE, "upload before attaching a communicator");
  if (int rc = set_device(c)) return rc;
  const int64_t U = transposed ? c->n : c->m, V = transposed ? c->m : c->n;   // the user's matrix is U x V
  if (indptr[0] != 0 || indptr[U] != nnz) return fail(NBMF_ERR_ARG, "indptr does not span the %lld stored entries", (long long)nnz);
  if (masked && (mask_indptr[0] != 0 || mask_indptr[U] != mask_nnz)) return fail(NBMF_ERR_ARG, "mask indptr does not span its entries");
  for (void** p : {&c->dataA, &c->dataB, &c->maskA, &c->maskB}) {
    if (*p) dfree(*p);
    *p = nullptr;
  }
  c->data_kind = -1;
  const long long RbA = c->mA / 16, RbB = c->nA / 16;
  const size_t tiles = (size_t)RbA * RbB;
  HIPCHK(dmalloc(&c->dataA, tiles * 256 + PASS_SLACK));
  HIPCHK(dmalloc(&c->dataB, tiles * 256 + PASS_SLACK));
  HIPCHK(hipMemsetAsync(c->stats, 0, sizeof(unsigned long long) * 8, c->stream));
  const long long dwords = (long long)tiles * 64;
  hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)((dwords + 255) / 256)), dim3(256), 0, c->stream, (uint32_t*)c->dataA,
                     (uint32_t*)c->dataB, RbA, RbB, (long long)c->m, (long long)c->n,
                     masked ? (unsigned)CB_VALID : (unsigned)(CB_VALID | CB_ZOBS));
  HIPCHK(hipGetLastError());
  // the two patterns, staged through the device in pieces of at most 64 M entries
  struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
      if (p) dfree(p);
    }
  } d_ptr, d_idx;
  HIPCHK(dmalloc(&d_ptr.p, sizeof(int64_t) * (size_t)(U + 1)));
  int64_t piece = 64ll << 20;
  if (const char* e = getenv("NBMF_CSR_PIECE")) piece = std::max<int64_t>(1, atoll(e));   // (tests: force several pieces)
  HIPCHK(dmalloc(&d_idx.p, sizeof(int32_t) * (size_t)std::max<int64_t>(1, std::min<int64_t>(piece, std::max(nnz, mask_nnz)))));
  for (int what = masked ? 0 : 1; what <= 1; ++what) {
    const int64_t* ip = what == 0 ? mask_indptr : indptr;
    const int32_t* ix = what == 0 ? mask_indices : indices;
    const int64_t cnt = what == 0 ? mask_nnz : nnz;
    HIPCHK(hipMemcpyAsync(d_ptr.p, ip, sizeof(int64_t) * (size_t)(U + 1), hipMemcpyHostToDevice, c->stream));
    for (int64_t e0 = 0; e0 < cnt; e0 += piece) {
      const int64_t ne = std::min(piece, cnt - e0);
      HIPCHK(hipMemcpyAsync(d_idx.p, ix + e0, sizeof(int32_t) * (size_t)ne, hipMemcpyHostToDevice, c->stream));
      // the kernel indexes entries from e0: shift the view of indices, keep indptr absolute
      hipLaunchKernelGGL(csr_scatter_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, c->stream,
                         (const long long*)d_ptr.p, (const int*)d_idx.p - e0, (long long)U, (long long)V, (long long)e0,
                         (long long)(e0 + ne), transposed, what, (unsigned char*)c->dataA, (unsigned char*)c->dataB, RbA, RbB, c->stats);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(c->stream));   // the staging buffer is reused by the next piece
    }
  }
  if (masked) {
    hipLaunchKernelGGL(count_observed_kernel, dim3(1024), dim3(256), 0, c->stream, (const uint32_t*)c->dataA, dwords, c->stats);
    HIPCHK(hipGetLastError());
  }
  unsigned long long st[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(st, c->stats, sizeof st, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (st[1] != 0) return fail(NBMF_ERR_ARG, "%llu column indices outside [0, %lld)", st[1], (long long)V);
  c->data_kind = DATA_BIN;
  if (int rc2 = setup_workspaces(c)) return rc2;
  c->n_obs = masked ? (double)st[0] : (double)c->m * (double)c->n;
  if (int rc2 = mark_pad_rows_b(c)) return rc2;
  c->n_obs_global = c->n_obs;
  hipLaunchKernelGGL(rowcount_kernel, dim3((unsigned)((c->m + 255) / 256)), dim3(256), 0, c->stream, c->dataB, c->maskB,
                     c->data_kind, (long long)(c->nA / 16), (long long)c->m, (long long)c->n, c->rowcnt);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  if (out_flags) *out_flags = NBMF_FLAG_BINARY_PATH;
  return NBMF_OK;
}

int nbmf_get_n_obs(nbmf_ctx* c, double* n_obs) {
  if (!c || !n_obs) return fail(NBMF_ERR_ARG, "null argument");
  if (c->data_kind < 0) return fail(NBMF_ERR_STATE, "nbmf_upload has not been called");
  *n_obs = c->n_obs;
  return NBMF_OK;
}

// Are the factors on the device where a fit keeps them (tiny_a)?  flags[5] collects the violations; synchronises.
int check_factor_range(nbmf_ctx* c) {
  HIPCHK(hipMemsetAsync(c->flags + 5, 0, sizeof(int), c->stream));
  const long long tot = std::max((long long)c->mA, (long long)c->KP * c->nA);
  hipLaunchKernelGGL(factor_range_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->Wn,
                     (const double*)c->Hn, c->k, (long long)c->m, (long long)c->mA, (long long)c->n, (long long)c->nA, c->flags + 5);
  HIPCHK(hipGetLastError());
  int out_of_range = 1;
  HIPCHK(hipMemcpyAsync(&out_of_range, c->flags + 5, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->factors_in_range = out_of_range == 0;
  return NBMF_OK;
}

int nbmf_set_factors(nbmf_ctx* c, const double* W, const double* H) {
  if (!c || !W || !H) return fail(NBMF_ERR_ARG, "null argument");
  if (int rc = set_device(c)) return rc;
  HIPCHK(hipMemcpyAsync(c->stage, W, sizeof(double) * (size_t)c->k * c->m, hipMemcpyHostToDevice, c->stream));
  {
    const long long tot = (long long)c->KP * c->mA;
    hipLaunchKernelGGL(set_factor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->stage, c->Wn,
                       c->WT, c->WG, c->k, c->KP, std::min(c->KP, SLICE_K), (long long)c->m, (long long)c->mA);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpyAsync(c->stage, H, sizeof(double) * (size_t)c->k * c->n, hipMemcpyHostToDevice, c->stream));
  {
    const long long tot = (long long)c->KP * c->nA;
    hipLaunchKernelGGL(set_factor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->stage, c->Hn,
                       c->HT, c->HG, c->k, c->KP, std::min(c->KP, SLICE_K), (long long)c->n, (long long)c->nA);
    HIPCHK(hipGetLastError());
  }
  hipLaunchKernelGGL(prior_kernel, dim3(c->n_prior_blocks), dim3(256), 0, c->stream, c->Hn, c->prior, c->k, c->KP,
                     (long long)c->n, (long long)c->nA, c->eps);
  HIPCHK(hipGetLastError());
  if (int rc = check_factor_range(c)) return rc;
  c->prior_src = c->prior;
  c->n_prior_src = c->n_prior_blocks;
  c->have_factors = true;
  return NBMF_OK;
}

int nbmf_get_factors(nbmf_ctx* c, double* W, double* H) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (!c->have_factors) return fail(NBMF_ERR_STATE, "no factors on the device");
  if (int rc = set_device(c)) return rc;
  if (W) {
    const long long tot = (long long)c->k * c->m;
    hipLaunchKernelGGL(get_factor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->Wn, c->stage,
                       c->k, (long long)c->m, (long long)c->mA);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(W, c->stage, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (H) {
    const long long tot = (long long)c->k * c->n;
    hipLaunchKernelGGL(get_factor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->Hn, c->stage,
                       c->k, (long long)c->n, (long long)c->nA);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(H, c->stage, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return NBMF_OK;
}

int nbmf_run(nbmf_ctx* c, int max_iter, double tol, double* losses, int* n_iter) {
  if (int rc = ready(c)) return rc;
  if (max_iter < 1) return fail(NBMF_ERR_ARG, "max_iter must be >= 1");
  if (!losses || !n_iter) return fail(NBMF_ERR_ARG, "null output");
  if (c->cancelled.load(std::memory_order_relaxed)) return fail(NBMF_ERR_STATE, "cancelled (nbmf_cancel)");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_losses(c, max_iter)) return rc;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->scal, 0, sizeof(double) * 8, c->stream));
  {
    bool handled = false;   // small problems: the whole run in one persistent launch
    if (int rc = run_small(c, max_iter, tol, losses, n_iter, &handled)) return rc;
    if (handled) return NBMF_OK;
  }
  g_engine_launches_served.fetch_add(1, std::memory_order_relaxed);

  // Timeline (N3 of SURVEY Appendix A): the H-pass of iteration t also yields the log-likelihood of
  // the factors after iteration t-1, so loss(t-1) and its stop test are settled before H-update(t).
  const bool progress = c->progress && c->progress_every > 0;
  const int batch = progress ? c->progress_every : (tol > 0.0) ? 8 : max_iter;
  int reported = 0;   // losses already handed to the progress callback
  std::vector<double> fresh;
  auto report = [&](int n_final) -> int {
    if (!progress || n_final <= reported) return NBMF_OK;
    fresh.resize((size_t)(n_final - reported));
    HIPCHK(hipMemcpy(fresh.data(), c->losses_d + reported, sizeof(double) * fresh.size(), hipMemcpyDeviceToHost));
    c->progress(c->progress_user, reported, n_final - reported, fresh.data());
    reported = n_final;
    return NBMF_OK;
  };
  int host_done = 0;
  int it = 0;
  // One iteration = five launches with identical arguments every time (the loss index is counted on the
  // device), so on a single GPU it can be captured once into a hipGraph and replayed.  Measured: no gain
  // (config 1, 100x500 K=6: 14.7k it/s replayed vs 16.5k it/s eager) -- tiny problems are bound by the
  // latency of the five dependent kernels, not by the host's launch rate -- so it is opt-in
  // (NBMF_USE_GRAPH=1), never used with a communicator or event timing.
  const bool use_graph = getenv("NBMF_USE_GRAPH") && !is_sharded(c) && !c->timing && !progress && max_iter >= 8;
  bool fused_fin = fin_fusable(c) && !use_graph;
  // A sweep whose loss assembly ran out of time (PassFin: the chip is shared and this sweep's other workgroups were held
  // back for seconds) has assembled nothing, raised the stop flag -- every later kernel has returned at once, the factors
  // are those that sweep read -- and told the host (flags[6]).  `done` losses are in place (flags[1]): the run resumes at
  // iteration done + 1, whose H sweep is the one that failed, with the loss and stop test in a launch of their own from
  // here on (bit for bit the fused form's losses: tested).  Slots that filled late are emptied first.
  auto recover = [&](int done) -> int {
    g_loss_assembly_recoveries.fetch_add(1, std::memory_order_relaxed);
    if (getenv("NBMF_DEBUG")) fprintf(stderr, "[nbmf] loss %d could not be assembled inside its sweep within the bound: resuming with separate launches\n", done);
    fused_fin = false;
    if (int rc = fill_ll_empty(c)) return rc;
    HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int), c->stream));
    HIPCHK(hipMemsetAsync(c->flags + 6, 0, sizeof(int), c->stream));
    return NBMF_OK;
  };
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  if (use_graph) {
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_h_pass(c);
    if (!rc) rc = enqueue_finalize(c, -1, tol);
    if (!rc) rc = enqueue_h_update(c);
    if (!rc) rc = enqueue_w_step(c, c->projection);
    hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc) return rc;
    if (e != hipSuccess) return fail(NBMF_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    HIPCHK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
  }
  struct GraphGuard {
    hipGraph_t g;
    hipGraphExec_t x;
    ~GraphGuard() {
      if (x) hipGraphExecDestroy(x);
      if (g) hipGraphDestroy(g);
    }
  } guard{graph, gexec};
  for (int recoveries = 0;; ++recoveries) {
  while (it < max_iter && !host_done) {
    const int end = std::min(max_iter, it + batch);
    for (; it < end; ++it) {
      if (c->cancelled.load(std::memory_order_relaxed)) return fail(NBMF_ERR_STATE, "cancelled (nbmf_cancel)");
      c->timing_it = it;
      if (use_graph) {
        HIPCHK(hipGraphLaunch(gexec, c->stream));
        continue;
      }
      if (is_sharded(c) && c->shard_axis == 0) {
        if (int rc = enqueue_iteration_rows(c, it, tol)) return rc;
        continue;
      }
      if (fused_fin) {
        if (int rc = enqueue_h_pass(c, it - 1, tol)) return rc;   // (with the loss and stop test of iteration it-1)
      } else {
        if (int rc = enqueue_h_pass(c)) return rc;
        if (it > 0)
          if (int rc = enqueue_finalize(c, it - 1, tol)) return rc;
      }
      if (int rc = enqueue_h_update(c)) return rc;
      if (int rc = enqueue_w_step(c, c->projection)) return rc;
    }
    if ((tol > 0.0 || progress) && it < max_iter) {
      int fl[8];
      HIPCHK(hipMemcpyAsync(fl, c->flags, sizeof fl, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      if (c->timing) timing_collect(c);
      if (int rc = peer_check(c)) return rc;
      if (fl[6] && fused_fin && recoveries < 4) {
        if (int rc = recover(fl[1])) return rc;
        it = fl[1] + 1;
        continue;
      }
      host_done = fl[0];
      if (int rc = report(fl[1])) return rc;
    }
  }
  if (!host_done) {
    // loss of the last iteration (Theta-only sweep)
    if (fused_fin) {
      if (int rc = enqueue_loglik_pass(c, 0, 0, max_iter - 1, tol)) return rc;
    } else {
      if (int rc = enqueue_loglik_pass(c, 0)) return rc;
      if (int rc = enqueue_finalize(c, use_graph ? -1 : max_iter - 1, tol)) return rc;
    }
  }
  int fl[8];
  HIPCHK(hipMemcpyAsync(fl, c->flags, sizeof fl, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->timing) timing_collect(c);
  if (int rc = peer_check(c)) return rc;
  if (fl[6]) {   // the loss assembly inside a sweep gave up waiting for a partial (PassFin)
    if (fused_fin && recoveries < 4 && !c->cancelled.load(std::memory_order_relaxed)) {
      if (int rc = recover(fl[1])) return rc;
      it = fl[1] + 1;
      host_done = 0;
      continue;   // the iterations from the one whose sweep failed on (or, behind the last one, only the last loss)
    }
    if (int rc = fill_ll_empty(c)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return fail(NBMF_ERR_STATE, "internal: a sweep's loss assembly timed out waiting for its workgroups' partials");
  }
  if (c->cancelled.load(std::memory_order_relaxed)) return fail(NBMF_ERR_STATE, "cancelled (nbmf_cancel)");
  const int nit = fl[1];
  if (nit < 1 || nit > max_iter) return fail(NBMF_ERR_STATE, "internal: device reported n_iter=%d", nit);
  if (int rc = report(nit)) return rc;
  HIPCHK(hipMemcpy(losses, c->losses_d, sizeof(double) * (size_t)nit, hipMemcpyDeviceToHost));
  *n_iter = nit;
  return NBMF_OK;
  }
}

int nbmf_run_batch(nbmf_ctx* c, int n_problems, const double* alpha, const double* beta, const double* W0, const double* H0,
                   int max_iter, double tol, double* losses, int* n_iter, double* W_out, double* H_out) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (c->data_kind < 0) return fail(NBMF_ERR_STATE, "nbmf_upload has not been called");
  if (n_problems < 1) return fail(NBMF_ERR_ARG, "n_problems must be >= 1");
  if (max_iter < 1) return fail(NBMF_ERR_ARG, "max_iter must be >= 1");
  if (!alpha || !beta || !W0 || !H0 || !losses || !n_iter || !W_out || !H_out) return fail(NBMF_ERR_ARG, "null argument");
  if (is_sharded(c)) return fail(NBMF_ERR_STATE, "nbmf_run_batch runs on an unsharded context (nbmf_comm_detach first)");
  if (int rc = set_device(c)) return rc;
  {
    bool handled = false;
    if (int rc = run_small_batch(c, n_problems, alpha, beta, W0, H0, max_iter, tol, losses, n_iter, W_out, H_out, &handled)) return rc;
    if (handled) return NBMF_OK;
  }
  // problems too large for the persistent kernel (or a run that gave up): one after the other, each exactly the
  // nbmf_set_hyper / nbmf_set_factors / nbmf_run / nbmf_get_factors sequence
  const double a0 = c->alpha, b0 = c->beta;
  const size_t wsz = (size_t)c->k * c->m, hsz = (size_t)c->k * c->n;
  int rc = NBMF_OK;
  for (int p = 0; p < n_problems && rc == NBMF_OK; ++p) {
    c->alpha = alpha[p];
    c->beta = beta[p];
    rc = nbmf_set_factors(c, W0 + (size_t)p * wsz, H0 + (size_t)p * hsz);
    if (rc == NBMF_OK) rc = nbmf_run(c, max_iter, tol, losses + (size_t)p * max_iter, n_iter + p);
    if (rc == NBMF_OK) rc = nbmf_get_factors(c, W_out + (size_t)p * wsz, H_out + (size_t)p * hsz);
  }
  c->alpha = a0;
  c->beta = b0;
  return rc;
}

int nbmf_batch_stats(nbmf_ctx* c, int* launches, int* problems) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (launches) *launches = c->small_batch.launches;
  if (problems) *problems = c->small_batch.problems;
  return NBMF_OK;
}

int nbmf_sweep_info(nbmf_ctx* c, int* h_chunks, int* h_blocks, int* w_chunks, int* w_blocks) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (!c->cstartH) return fail(NBMF_ERR_STATE, "no data on the device yet");
  if (h_chunks) *h_chunks = c->chunksH;
  if (h_blocks) *h_blocks = c->CH_H;
  if (w_chunks) *w_chunks = c->chunksW;
  if (w_blocks) *w_blocks = c->CH_W;
  return NBMF_OK;
}

int nbmf_variant_stats(long long* full_w_launches, long long* ragged_k_launches, long long* loss_assembly_recoveries) {
  if (full_w_launches) *full_w_launches = g_full_w_launches.load(std::memory_order_relaxed);
  if (ragged_k_launches) *ragged_k_launches = g_ragged_launches.load(std::memory_order_relaxed);
  if (loss_assembly_recoveries) *loss_assembly_recoveries = g_loss_assembly_recoveries.load(std::memory_order_relaxed);
  return NBMF_OK;
}

int nbmf_w_only_steps(nbmf_ctx* c, int n_steps) {
  if (int rc = ready(c)) return rc;
  if (n_steps < 0) return fail(NBMF_ERR_ARG, "n_steps must be >= 0");
  if (int rc = set_device(c)) return rc;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));
  struct FreeW {
    nbmf_ctx* c;
    ~FreeW() { c->w_free = false; }
  } free_w{c};
  c->w_free = true;   // (W starts off the simplex, _base.py:175: Theta above 1 is possible)
  for (int s = 0; s < n_steps; ++s) {
    if (int rc = enqueue_w_step(c, NBMF_PROJ_NORMALIZE)) return rc;
    if (s == 0 && n_steps > 2) {
      // ... and is on it from the first step on: looked at once (a ~20 us kernel and a synchronisation), so that the
      // remaining steps can take the sweeps' plain variant if W has come out non-negative as well
      if (int rc = check_factor_range(c)) return rc;
      c->w_free = !c->factors_in_range;
    }
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->timing) timing_collect(c);
  return peer_check(c);
}

int nbmf_loss(nbmf_ctx* c, double* loss) {
  if (int rc = ready(c)) return rc;
  if (!loss) return fail(NBMF_ERR_ARG, "null output");
  return loss_by_launches(c, loss);
}
int nbmf_loglik(nbmf_ctx* c, int clip_theta, double* loglik) {
  if (int rc = ready(c)) return rc;
  if (!loglik) return fail(NBMF_ERR_ARG, "null output");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_losses(c, 1)) return rc;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));
  if (int rc = enqueue_loglik_pass(c, 0, clip_theta != 0)) return rc;
  if (int rc = enqueue_finalize(c, 0, 0.0, /*loglik_only=*/true)) return rc;   // -(ll + 0 + 0) / -1 = ll
  HIPCHK(hipMemcpyAsync(loglik, c->losses_d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->timing) timing_collect(c);
  return peer_check(c);
}

int nbmf_loglik_strict(nbmf_ctx* c, double* loglik) {
  if (int rc = ready(c)) return rc;
  if (!loglik) return fail(NBMF_ERR_ARG, "null output");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_losses(c, 1)) return rc;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));
  if (int rc = enqueue_loglik_pass(c, 1)) return rc;
  if (int rc = enqueue_finalize(c, 0, 0.0, /*loglik_only=*/true, /*strict=*/1)) return rc;   // no pad term, no prior
  HIPCHK(hipMemcpyAsync(loglik, c->losses_d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return peer_check(c);
}

int nbmf_comm_unique_id(void* id128) {
  if (!id128) return fail(NBMF_ERR_ARG, "null id");
  if (int rc = load_rccl()) return rc;
  NCCLCHK(g_rccl.GetUniqueId(id128));
  return NBMF_OK;
}

// Shared tail of the two comm-init entry points: exchange buffers for the chosen axis and the global
// quantities the update formulas need.
static int comm_finish_init(nbmf_ctx* c, int nranks, int rank, int shard_axis) {
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));   // a stop flag left by an earlier run must not mute the exchanges below
  c->nranks = nranks;
  c->rank = rank;
  c->shard_axis = shard_axis;
  // row split: panels of the K x N exchange (see enqueue_iteration_rows).  Two panels (NBMF_OVERLAP=1)
  // hide half of the exchange but cost ~35 us per iteration in extra launches and a split W-pass (measured
  // with one rank, where there is nothing to hide); the break-even is an all-reduce of ~70 us, which cannot
  // be timed on a one-GPU box, so the default stays one panel.
  const char* ov = getenv("NBMF_OVERLAP");
  const bool two = c->panels_wanted ? c->panels_wanted == 2 : (ov && atoi(ov) != 0);
  c->npanel = (shard_axis == 0 && c->KS == 1 && c->chunksW >= 2 && two) ? 2 : 1;
  c->wsplit = c->npanel == 2 ? c->chunksW / 2 : c->chunksW;
  c->pc0[0] = 0;
  c->pc0[1] = c->npanel == 2 ? (long long)c->bW_host[c->wsplit] * 16 : c->nA;
  c->pc0[2] = c->nA;
  c->pbase[0] = 0;
  c->ll_index = 2 * (size_t)c->KP * c->pc0[1];
  c->pbase[1] = c->ll_index + 2;
  if (getenv("NBMF_DEBUG"))
    fprintf(stderr, "[nbmf] rank %d/%d axis %d: %d exchange panel(s), columns split at %lld of %lld, W-pass chunks %d + %d, %s\n",
            rank, nranks, shard_axis, c->npanel, c->pc0[1], (long long)c->nA, c->wsplit, c->chunksW - c->wsplit,
            c->comm ? "RCCL" : c->peer ? "peer (xGMI)" : "host transport");
  if (c->npanel == 2 && (c->comm || c->peer) && !c->stream2) {
    HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->evH, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->evF, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev1, hipEventDisableTiming));
  }
  if (c->peer && shard_axis == 0) {
    // the slice of each panel's columns this rank reduces and updates, in 16-column blocks
    for (int p = 0; p < c->npanel; ++p) {
      const long long b_lo = c->pc0[p] / 16, nb = (c->pc0[p + 1] - c->pc0[p]) / 16;
      const long long b0 = nb * rank / nranks, b1 = nb * (rank + 1) / nranks;
      c->psl_c0[p] = 16 * (b_lo + b0);
      c->psl_wp[p] = 16 * (b1 - b0);
    }
  }
  if (shard_axis == 1 && !c->Qbuf) HIPCHK(dmalloc(&c->Qbuf, sizeof(double) * (size_t)c->KP * c->mA));
  // global observed count (the divisor of _solver.py:162) and, when the columns are split, the global
  // column count (the "/ n" of :54) and the per-row observed counts of the Duchi extension
  double h[2] = {c->n_obs, (double)c->n};
  HIPCHK(hipMemcpyAsync(c->sbuf, h, sizeof h, hipMemcpyHostToDevice, c->stream));
  if (int rc = all_reduce_inplace(c, c->sbuf, 2)) return rc;
  HIPCHK(hipMemcpyAsync(h, c->sbuf, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->n_obs_global = h[0];
  c->n_div_global = (shard_axis == 1) ? h[1] : (double)c->n;
  if (shard_axis == 1) {
    HIPCHK(hipMemsetAsync(c->rowcnt + c->m, 0, sizeof(double) * (size_t)(c->mA - c->m), c->stream));
    if (c->peer) {   // the peer exchange works on the arena only
      HIPCHK(hipMemcpyAsync(c->arena, c->rowcnt, sizeof(double) * (size_t)c->mA, hipMemcpyDeviceToDevice, c->stream));
      if (int rc = all_reduce_inplace(c, c->arena, (size_t)c->mA)) return rc;
      HIPCHK(hipMemcpyAsync(c->rowcnt, c->arena, sizeof(double) * (size_t)c->mA, hipMemcpyDeviceToDevice, c->stream));
    } else {
      if (int rc = all_reduce_inplace(c, c->rowcnt, (size_t)c->mA)) return rc;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return NBMF_OK;
}

static int comm_check_args(nbmf_ctx* c, int nranks, int rank, int shard_axis) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(NBMF_ERR_ARG, "bad rank %d / nranks %d", rank, nranks);
  if (shard_axis != 0 && shard_axis != 1) return fail(NBMF_ERR_ARG, "shard_axis must be 0 (rows of Y) or 1 (columns of Y)");
  if (c->data_kind < 0) return fail(NBMF_ERR_STATE, "call nbmf_upload before attaching a communicator (global counts are reduced there)");
  if (is_sharded(c)) return fail(NBMF_ERR_STATE, "a communicator is already attached (nbmf_comm_detach first)");
  return NBMF_OK;
}

int nbmf_comm_init(nbmf_ctx* c, const void* id128, int nranks, int rank, int shard_axis) {
  if (!c || !id128) return fail(NBMF_ERR_ARG, "null argument");
  if (int rc = comm_check_args(c, nranks, rank, shard_axis)) return rc;
  if (int rc = set_device(c)) return rc;
  if (int rc = load_rccl()) return rc;
  Uid uid;
  memcpy(uid.internal, id128, 128);
  NCCLCHK(g_rccl.CommInitRank(&c->comm, nranks, uid, rank));
  return comm_finish_init(c, nranks, rank, shard_axis);
}

int nbmf_comm_init_host(nbmf_ctx* c, nbmf_host_allreduce_fn fn, void* user, int nranks, int rank, int shard_axis) {
  if (!c || !fn) return fail(NBMF_ERR_ARG, "null argument");
  if (int rc = comm_check_args(c, nranks, rank, shard_axis)) return rc;
  if (int rc = set_device(c)) return rc;
  const size_t cnt = std::max<size_t>(2 * (size_t)c->KP * c->nA + 1, (size_t)c->KP * c->mA);
  if (!c->host_buf) {
    HIPCHK(hipHostMalloc((void**)&c->host_buf, cnt * sizeof(double), hipHostMallocDefault));
    c->host_buf_count = cnt;
  }
  c->host_reduce = fn;
  c->host_reduce_user = user;
  return comm_finish_init(c, nranks, rank, shard_axis);
}

// Cooperative cancellation from ANOTHER thread (the only entry point that may be called on a context while its owner is
// inside a call): the flag is sticky; what is already enqueued is cut short on the device by raising the run's own stop
// flag -- the word every kernel of an iteration (the exchange kernels included) looks at before it does anything, as after
// convergence -- through a stream of its own.  The owner's nbmf_run then returns NBMF_ERR_STATE ("cancelled").
int nbmf_cancel(nbmf_ctx* c) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  c->cancelled.store(1, std::memory_order_relaxed);
  int* flags = c->flags;
  if (!flags) return NBMF_OK;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = nullptr;
  HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  static const int one = 1;
  hipError_t e = hipMemcpyAsync(flags, &one, sizeof(int), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipStreamDestroy(st);
  if (e != hipSuccess) return fail(NBMF_ERR_HIP, "nbmf_cancel: %s", hipGetErrorString(e));
  return NBMF_OK;
}

int nbmf_set_peer_timeout_ms(nbmf_ctx* c, double ms) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (!(ms >= 0.0)) return fail(NBMF_ERR_ARG, "timeout must be >= 0 ms");
  c->peer_timeout_ms = ms;
  return NBMF_OK;
}

int nbmf_set_exchange_panels(nbmf_ctx* c, int panels) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (panels < 0 || panels > 2) return fail(NBMF_ERR_ARG, "panels must be 0 (default), 1 or 2");
  c->panels_wanted = panels;
  return NBMF_OK;
}

int nbmf_comm_detach(nbmf_ctx* c) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (int rc = set_device(c)) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->stream2) HIPCHK(hipStreamSynchronize(c->stream2));
  comm_release(c, /*recount=*/true);
  return NBMF_OK;
}

// What the attached transport ITSELF says about the job (as opposed to what the caller told it): for RCCL the
// communicator's own rank count and device (ncclCommCount / ncclCommCuDevice: -1 where this librccl lacks them), for the
// peer transport the number of arenas it has mapped (its own included) and how many of those it reaches over IPC or peer
// access -- i.e. outside this context's own device memory.
int nbmf_comm_info(nbmf_ctx* c, int* kind, int* nranks_seen, int* remote) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  int k = 0, n = 0, rem = 0;
  if (c->comm) {
    k = 1;
    n = -1;
    rem = -1;
    if (g_rccl.CommCount) NCCLCHK(g_rccl.CommCount(c->comm, &n));
    if (g_rccl.CommCuDevice) NCCLCHK(g_rccl.CommCuDevice(c->comm, &rem));
  } else if (c->peer) {
    k = 2;
    for (int j = 0; j < c->pv.nranks; ++j)
      if (c->pv.arena[j] && c->pv.flag[j]) {
        ++n;
        if (j != c->pv.rank) ++rem;
      }
  } else if (c->host_reduce) {
    k = 3;
    n = c->nranks;
    rem = c->nranks - 1;
  }
  if (kind) *kind = k;
  if (nranks_seen) *nranks_seen = n;
  if (remote) *remote = rem;
  return NBMF_OK;
}

// Arena layout in doubles (same on every rank for a given axis, because the unsplit extent is):
//   axis 0   X = [P1 | P2 | loglik .. 8]   HX = H' staging [KP][nA]   PR = 2 x prior partial pairs   SC = 64 scalars
//   axis 1   X = [Q (KP x mA) .. 8]        SC = 64 scalars
static void peer_layout(nbmf_ctx* c, int axis) {
  const long long fh = (long long)c->KP * c->nA, fw = (long long)c->KP * c->mA;
  if (axis == 0) {
    c->x_doubles = 2 * fh + 8;
    c->offHX = c->x_doubles;
    c->offPR = c->offHX + fh;
    c->offSC = c->offPR + 2LL * 2 * 2 * PEER_H_WGS * PEER_MAX_RANKS;   // [parity][panel][workgroup of every rank][2]
  } else {
    c->x_doubles = fw + 8;
    c->offHX = c->offPR = c->offSC = c->x_doubles;
  }
}

int nbmf_peer_export(nbmf_ctx* c, int shard_axis, void* handle) {
  if (!c || !handle) return fail(NBMF_ERR_ARG, "null argument");
  if (shard_axis != 0 && shard_axis != 1) return fail(NBMF_ERR_ARG, "shard_axis must be 0 or 1");
  if (is_sharded(c)) return fail(NBMF_ERR_STATE, "a communicator is already attached (nbmf_comm_detach first)");
  if (int rc = set_device(c)) return rc;
  // Epochs restart at zero on all ranks together: arena and flag block are cleared on every export (the
  // ranks synchronise on the handle exchange before anyone signals).  The memory comes from the process-wide
  // pool and keeps its IPC handles for the life of the process.
  if (!c->arena || c->arena_axis != shard_axis) {
    arena_release((ArenaSlot*)c->arena_slot.p);
    c->arena_slot.p = nullptr;
    c->arena = nullptr;
    c->pflags = nullptr;
    peer_layout(c, shard_axis);
    ArenaSlot* slot = nullptr;
    if (int rc = arena_acquire(c->device, (size_t)(c->offSC + 64), &slot)) return rc;
    c->arena_slot.p = slot;
    c->arena = slot->arena;
    c->pflags = slot->flags;
    c->arena_doubles = slot->doubles;
    c->arena_axis = shard_axis;
  }
  HIPCHK(hipMemset(c->arena, 0, c->arena_doubles * sizeof(double)));
  HIPCHK(hipMemset(c->pflags, 0, PF_WORDS * sizeof(unsigned long long)));
  HIPCHK(hipDeviceSynchronize());
  c->epoch = 0;
  c->hseq = 0;
  memcpy(handle, ((ArenaSlot*)c->arena_slot.p)->handles, NBMF_PEER_HANDLE_BYTES);
  return NBMF_OK;
}

int nbmf_comm_init_peer(nbmf_ctx* c, const void* handles, int nranks, int rank, int shard_axis) {
  if (!c || !handles) return fail(NBMF_ERR_ARG, "null argument");
  if (int rc = comm_check_args(c, nranks, rank, shard_axis)) return rc;
  if (nranks > PEER_MAX_RANKS) return fail(NBMF_ERR_ARG, "the peer transport supports at most %d ranks", PEER_MAX_RANKS);
  if (!c->arena || c->arena_axis != shard_axis)
    return fail(NBMF_ERR_STATE, "call nbmf_peer_export(ctx, %d, ...) on every rank first", shard_axis);
  if (shard_axis == 0 && c->nA / 16 < nranks) return fail(NBMF_ERR_ARG, "too few columns to slice the H-step over %d ranks", nranks);
  if (int rc = set_device(c)) return rc;
  PeerView pv{};
  pv.nranks = nranks;
  pv.rank = rank;
  double ms = 30000.0;
  if (const char* e = getenv("NBMF_PEER_TIMEOUT_MS")) ms = std::max(1.0, atof(e));
  if (c->peer_timeout_ms > 0) ms = c->peer_timeout_ms;
  pv.timeout = (unsigned long long)(ms * 1e5);   // wall_clock64 ticks at 100 MHz
  struct Undo {   // any failure below leaves the context unattached
    nbmf_ctx* c;
    bool armed = true;
    ~Undo() {
      if (armed) {
        hipStreamSynchronize(c->stream);
        comm_release(c, /*recount=*/true);
      }
    }
  } undo{c};
  for (int j = 0; j < nranks; ++j) {
    if (j == rank) {
      pv.arena[j] = c->arena;
      pv.flag[j] = c->pflags;
      continue;
    }
    hipIpcMemHandle_t h[2];
    PeerOrigin origin;
    memcpy(h, (const char*)handles + (size_t)j * NBMF_PEER_HANDLE_BYTES, sizeof h);
    memcpy(&origin, (const char*)handles + (size_t)j * NBMF_PEER_HANDLE_BYTES + sizeof h, sizeof origin);
    if (origin.pid == process_nonce()) {
      // a rank of THIS process (several contexts, one host thread each): its arena is plain device memory here;
      // if it lives on another GPU of this process, peer access makes it addressable (pooled arenas are never freed,
      // so the address stays valid for the life of the process)
      if (origin.device != c->device) {
        const hipError_t e = hipDeviceEnablePeerAccess(origin.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          return fail(NBMF_ERR_HIP, "hipDeviceEnablePeerAccess(%d) from device %d: %s", origin.device, c->device, hipGetErrorString(e));
        (void)hipGetLastError();
      }
      pv.arena[j] = (double*)origin.arena;
      pv.flag[j] = (unsigned long long*)origin.flags;
      continue;
    }
    void *a = nullptr, *f = nullptr;
    HIPCHK(hipIpcOpenMemHandle(&a, h[0], hipIpcMemLazyEnablePeerAccess));
    c->peer_mapped.push_back(a);
    HIPCHK(hipIpcOpenMemHandle(&f, h[1], hipIpcMemLazyEnablePeerAccess));
    c->peer_mapped.push_back(f);
    pv.arena[j] = (double*)a;
    pv.flag[j] = (unsigned long long*)f;
  }
  c->pv = pv;
  c->peer = true;
  c->Pbuf_own = c->Pbuf;
  c->sbuf_own = c->sbuf;
  c->Qbuf_own = c->Qbuf;
  c->Pbuf = c->arena;
  c->Qbuf = c->arena;
  c->sbuf = c->arena + c->offSC;
  if (shard_axis == 0) {
    // column slices of the fused H-step, in 16-column blocks
    const long long nb = c->nA / 16;
    const long long b0 = nb * rank / nranks, b1 = nb * (rank + 1) / nranks;
    c->sl_c0 = 16 * b0;
    c->sl_wp = 16 * (b1 - b0);
  }
  // Known-answer exchanges before anything depends on the transport (also the first rendezvous of the ranks):
  // rank r contributes (r+1)*((i+7t)%1021+1)+t, all exact in binary64; FOUR epochs are queued back to back, each
  // filling, reducing and checking on the device with no host synchronisation in between, so the arena and the
  // flag words are reused exactly as consecutive iterations of a run reuse them.
  {
    const long long cnt = std::min<long long>(c->x_doubles, 1 << 20);
    const unsigned long long full = c->pv.timeout;
    c->pv.timeout = std::min<unsigned long long>(full, 10ull * 100000000ull);
    unsigned long long* bad = c->stats + 4;
    // (the exchange kernels return at once while the run's stop flag is up -- and a run that ended by its stop rule
    //  leaves it up: clear it, or the known-answer epochs of a re-attach find nothing exchanged)
    HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));
    HIPCHK(hipMemsetAsync(bad, 0, sizeof(unsigned long long), c->stream));
    for (int t = 0; t < 4; ++t) {
      hipLaunchKernelGGL(peer_selftest_fill_kernel, dim3(256), dim3(256), 0, c->stream, c->arena, cnt, rank, t);
      HIPCHK(hipGetLastError());
      if (int rc = all_reduce_inplace(c, c->arena, (size_t)cnt)) return rc;
      hipLaunchKernelGGL(peer_selftest_check_kernel, dim3(256), dim3(256), 0, c->stream, (const double*)c->arena, cnt, nranks, t, bad);
      HIPCHK(hipGetLastError());
    }
    unsigned long long n_bad = 0;
    HIPCHK(hipMemcpyAsync(&n_bad, bad, sizeof n_bad, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (int rc = peer_check(c)) return rc;
    if (n_bad) return fail(NBMF_ERR_COMM, "peer transport self-test failed: %llu wrong elements over four back-to-back exchanges", n_bad);
    c->pv.timeout = full;
    HIPCHK(hipMemsetAsync(c->arena, 0, (size_t)cnt * sizeof(double), c->stream));
  }
  if (int rc = comm_finish_init(c, nranks, rank, shard_axis)) return rc;
  if (int rc = peer_check(c)) return rc;
  undo.armed = false;
  return NBMF_OK;
}

int nbmf_timing_enable(nbmf_ctx* c, int enable) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  c->timing = enable != 0;
  c->timing_stride = enable > 1 ? enable : 1;
  c->timing_it = 0;
  c->ev_used = 0;
  c->t_ms[0] = c->t_ms[1] = 0;
  c->t_n[0] = c->t_n[1] = 0;
  return NBMF_OK;
}

int nbmf_timing_get(nbmf_ctx* c, double* hpass_ms, int* hpass_launches, double* wpass_ms, int* wpass_launches) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (hpass_ms) *hpass_ms = c->t_ms[0];
  if (hpass_launches) *hpass_launches = c->t_n[0];
  if (wpass_ms) *wpass_ms = c->t_ms[1];
  if (wpass_launches) *wpass_launches = c->t_n[1];
  return NBMF_OK;
}

int nbmf_set_progress(nbmf_ctx* c, nbmf_progress_fn fn, void* user, int every) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (fn && every < 1) return fail(NBMF_ERR_ARG, "every must be >= 1");
  c->progress = fn;
  c->progress_user = user;
  c->progress_every = fn ? every : 0;
  return NBMF_OK;
}

int nbmf_small_stats(nbmf_ctx* c, int* runs, int* aborted) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (runs) *runs = c->small.runs;
  if (aborted) *aborted = c->small.aborted;
  return NBMF_OK;
}

int nbmf_device_synchronize(int device) {
  if (device < 0) return fail(NBMF_ERR_ARG, "device index must be >= 0");   // (before anything touches the runtime)
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipDeviceSynchronize());
  return NBMF_OK;
}

int nbmf_synchronize(nbmf_ctx* c) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (int rc = set_device(c)) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return NBMF_OK;
}

int nbmf_selftest_unary(int device, int op, int n, const double* x, double* y) {
  if (!x || !y || n < 1 || op < 0 || op > 7) return fail(NBMF_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(device));
  const double2* ltab10 = nullptr;
  HIPCHK(log_table_build(device, nullptr));   // (a self-test entry, no context: the null stream)
  HIPCHK(log_table_device(&ltab10));
  double *d = nullptr, *o = nullptr;
  HIPCHK(dmalloc(&d, sizeof(double) * (size_t)n));
  HIPCHK(dmalloc(&o, sizeof(double) * (size_t)n));
  HIPCHK(hipMemcpy(d, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(unary_test_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, d, o, n, ltab10);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(y, o, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  dfree(d);
  dfree(o);
  return NBMF_OK;
}

int nbmf_engine_stats(long long* persistent_served, long long* persistent_aborted, long long* launches_served) {
  if (persistent_served) *persistent_served = g_engine_persistent_served.load(std::memory_order_relaxed);
  // (a fit declined because the context's kernel gave up earlier counts as given up: it ran on the other engine)
  if (persistent_aborted)
    *persistent_aborted = g_engine_persistent_aborted.load(std::memory_order_relaxed) + g_engine_persistent_declined.load(std::memory_order_relaxed);
  if (launches_served) *launches_served = g_engine_launches_served.load(std::memory_order_relaxed);
  return NBMF_OK;
}

int nbmf_selftest_mfma_peak(int device, double target_ms, double* tflops, double* cycles_per_mfma, double* launch_ms) {
  if (!(target_ms > 0.0 && target_ms <= 2000.0)) return fail(NBMF_ERR_ARG, "target_ms must be in (0, 2000]");
  HIPCHK(hipSetDevice(device));
  DevInfo di;
  HIPCHK(device_info(device, &di));
  if (!di.gfx950) return fail(NBMF_ERR_HIP, "device %d is %s, not gfx950", device, di.arch);
  const int wgs = di.cus * 2;   // 4 waves each: two waves per SIMD (one does as well: the pipe is the limit)
  double* out = nullptr;
  HIPCHK(dmalloc(&out, sizeof(double) * (size_t)wgs * 256));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct Cleanup {   // (every return below, the HIPCHK ones included, frees what has been made)
    double*& out;
    hipEvent_t &e0, &e1;
    ~Cleanup() {
      if (e0) hipEventDestroy(e0);
      if (e1) hipEventDestroy(e1);
      if (out) dfree(out);
    }
  } cleanup{out, e0, e1};
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  auto timed = [&](int iters, float* ms) -> hipError_t {
    hipError_t e;
    if ((e = hipEventRecord(e0, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipEventRecord(e1, 0)) != hipSuccess) return e;
    if ((e = hipEventSynchronize(e1)) != hipSuccess) return e;
    return hipEventElapsedTime(ms, e0, e1);
  };
  // a trip is 8 MFMAs of 64 cycles on each of two waves of a SIMD: ~0.43 us at 2.4 GHz.  Warm up and calibrate with a
  // short launch, then one launch of about target_ms (long enough for the power management to act).
  float ms = 0;
  hipError_t e = timed(2000, &ms);
  if (e == hipSuccess) e = timed(2000, &ms);
  int iters = 2000;
  if (e == hipSuccess) {
    iters = (int)std::min(4.0e6, std::max(2000.0, 2000.0 * target_ms / std::max((double)ms, 1e-3)));
    e = timed(iters, &ms);
  }
  if (e != hipSuccess) return fail(NBMF_ERR_HIP, "MFMA peak self-test failed: %s", hipGetErrorString(e));
  const double mfmas_per_simd = 2.0 * 8.0 * (double)iters;                    // two waves per SIMD
  const double flop = (double)wgs * 4.0 * 8.0 * (double)iters * 2048.0;       // 16 x 16 x 4 multiply-adds per MFMA and wave
  if (tflops) *tflops = flop / ((double)ms * 1e-3) * 1e-12;
  if (cycles_per_mfma) *cycles_per_mfma = (double)ms * 1e-3 * 2.4e9 / mfmas_per_simd;
  if (launch_ms) *launch_ms = ms;
  return NBMF_OK;
}

}  // extern "C"

