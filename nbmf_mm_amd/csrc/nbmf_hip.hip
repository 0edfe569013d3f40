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

#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nbmf_hip.h"

#ifndef NBMF_DUAL_THETA
#define NBMF_DUAL_THETA 1   // two interleaved Theta accumulation chains where a wave has its SIMD to itself
#endif
#ifndef NBMF_STAGE_HALF
#define NBMF_STAGE_HALF 1   // 16 KiB stages (3 workgroups per CU at K <= 64); 0 = 32 KiB stages
#endif
#ifndef NBMF_LDS_DMA
#define NBMF_LDS_DMA 1   // stage the factor panels with global_load_lds (LDS-DMA); 0 = through registers
#endif

namespace {

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
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
enum { MODE_H = 0, MODE_W = 1, MODE_L = 2 };

// code bits of the binary path
enum : unsigned { CB_YM = 1u, CB_ZOBS = 2u, CB_VALID = 4u };

constexpr int PAD = 128;          // m and n are padded to multiples of 128 (8 row blocks, 8 strips)
constexpr int WG_WAVES = 4;       // waves (= column strips) per workgroup of the pass kernel
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

// Natural logarithm for the general (real-valued / weighted) path: ~35 VALU instructions instead of the
// ~75 of the library routine, which matters because f64 VALU work shares the pipe with the f64 MFMAs.
// Classic reduction x = 2^k (1+f), sqrt(2)/2 <= 1+f < sqrt(2); s = f/(2+f); log(1+f) = f - f^2/2 +
// s (f^2/2 + R(s^2)) with the degree-7 minimax R of Sun's fdlibm e_log.c (error < 1 ulp); the quotient
// uses the Newton reciprocal above.  Non-positive, non-finite and subnormal arguments take the library
// routine (same NaN / -inf results as NumPy).
__device__ __forceinline__ double log_fast(double x) {
  if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return log(x);
  int k;
  double m = frexp(x, &k);                       // m in [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  k = lo ? k - 1 : k;
  const double f = m - 1.0;
  const double s = f * rcp_nr(2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                                      2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  return dk * 6.93147180369123816490e-01 - ((hfsq - __builtin_fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}

__device__ __forceinline__ double wave_sum(double v) {
  // fixed butterfly order -> bitwise reproducible
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------------------------
// The fused pass kernel.
//   D      data matrix of this pass, R row blocks x C column strips of 16x16 tiles
//   L      the factor indexed by D's rows (streamed through LDS):  LT = Theta operand, LG = back operand
//   Rf     the factor indexed by D's columns (stationary in registers, T form)
//   MODE_H: out1 = L (R1), out2 = L (R2)                  (P1, P2 of _solver.py:42-43)
//   MODE_W: out1 = L (S1 - S2) + column sums of S2         (the bracket of _solver.py:53)
//   MODE_L: no products; strictly masked log-likelihood only (held-out perplexity, evaluation)
// Workgroup = 4 waves; wave w owns column strip 4*blockIdx.x + w and sweeps row blocks
// [chunk*CH, chunk*CH + CH).  Partial results go to per-chunk slabs (ordered reduction later: no
// atomics, bitwise reproducible).
// ------------------------------------------------------------------------------------------
struct PassArgs {
  const void* data;     // BIN: uint32 [C strips][Rb][64] ; F64: double [C strips][Rb][64][4]
  const void* mask;     // F64M only, same indexing as data
  const double* LT;     // [Rb][K/4][64]
  const double* LG;     // [Rb][K/16][4][64]
  const double* RfT;    // [Cb][K/4][64]
  double* out1;         // [chunks][K][C_alloc]
  double* out2;         // MODE_H only
  double* lossbuf;      // MODE_H/L: [chunks][Cb/4] per-workgroup log-likelihood partials
  const int* done;      // device stop flag: skip all work when set
  int Rb;               // row blocks of D (multiple of 8)
  int Cb;               // column strips of D (multiple of 4)
  const int* chunk_start;   // [chunks + 1] row-block boundaries of the chunks (long chunks first, short ones last)
  long long C_alloc;    // 16*Cb
  double eps;
  int strict;           // MODE_L: 1 = only observed entries enter the likelihood; 0 = the loss's semantics (as MODE_H)
  int clip;             // MODE_L: 1 = clip Theta to [0, 1] first (NBMFMM.inverse_transform, _base.py:210)
};

template <int KB, int DATA, int MODE>
__global__ __launch_bounds__(256, (NBMF_STAGE_HALF && KB <= 4) ? 3 : ((KB <= 4 || MODE == MODE_W) ? 2 : 1)) void pass_kernel(PassArgs a) {
  constexpr int K = 16 * KB;
  constexpr int S = K / 4;            // Theta k-steps
#if NBMF_STAGE_HALF
  constexpr int NB = (KB >= 4) ? 1 : 4 / KB;   // experiment: 16 KiB stages -> 3 workgroups per CU
#else
  constexpr int NB = 8 / KB;          // row blocks per LDS stage (32 KiB)
#endif
  constexpr int BLK = K * 16;         // doubles per block per operand image
  constexpr bool U8 = (NB * BLK / 2) / 256 == 8;   // 16-byte pieces per thread per image: 4, or 8 at K = 128
  constexpr int STAGE_D = 2 * NB * BLK;   // doubles per stage: [NB][T image] then [NB][G image] (= 32 KB)
  constexpr int N2 = NB * BLK / 2;    // double2 per image per stage (1024, or 2048 at K = 128)
  extern __shared__ __attribute__((aligned(16))) double lds[];   // two stages (double buffer)

  if (*a.done) return;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int cb = blockIdx.x * WG_WAVES + wave;
  const int chunk = blockIdx.y;
  const int rb0 = a.chunk_start[chunk];
  const int rb1 = a.chunk_start[chunk + 1];
  const double eps = a.eps;

  // stationary operand: Rf in T form for this strip
  double rf[S];
#pragma unroll
  for (int s = 0; s < S; ++s) rf[s] = a.RfT[((size_t)cb * S + s) * 64 + lane];

  d4 acc1[KB], acc2[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    acc1[i] = d4{0, 0, 0, 0};
    acc2[i] = d4{0, 0, 0, 0};
  }
  double prod = 1.0;   // BIN loss: running product of the per-entry Bernoulli probabilities
  int pexp = 0;        //           and its binary exponent
  double llsum = 0.0;  // F64 loss
  double s2 = 0.0;     // MODE_W: column sums of S2

  const uint32_t* codes = (const uint32_t*)a.data + (size_t)cb * a.Rb * 64 + lane;
  const d4* yv = (const d4*)a.data + (size_t)cb * a.Rb * 64 + lane;
  const d4* mv = (const d4*)a.mask + (size_t)cb * a.Rb * 64 + lane;

  // ---- staging: the operand images of NB row blocks are contiguous in HBM in exactly the LDS order,
  //      so a stage is a linear 2 x 16 KB copy; loads are issued one stage ahead (registers), written
  //      to the other LDS buffer after the current stage's math, one barrier per stage.
  // (macros over named scalars, not lambdas/arrays: hipcc leaves such an array in scratch memory once
  //  register pressure rises)
  double2 sr0, sr1, sr2, sr3, sr4, sr5, sr6, sr7, sr8, sr9, sr10, sr11, sr12, sr13, sr14, sr15;
  uint32_t cnext[NB];
#define STAGE_LOAD(RB)                                                          \
  {                                                                             \
    const double2* srcT_ = (const double2*)(a.LT + (size_t)(RB) * BLK) + threadIdx.x; \
    const double2* srcG_ = (const double2*)(a.LG + (size_t)(RB) * BLK) + threadIdx.x; \
    sr0 = srcT_[0];                                                             \
    sr1 = srcT_[256];                                                           \
    sr2 = srcT_[512];                                                           \
    sr3 = srcT_[768];                                                           \
    sr4 = srcG_[0];                                                             \
    sr5 = srcG_[256];                                                           \
    sr6 = srcG_[512];                                                           \
    sr7 = srcG_[768];                                                           \
    if (U8) {                                                                   \
      sr8 = srcT_[1024];                                                        \
      sr9 = srcT_[1280];                                                        \
      sr10 = srcT_[1536];                                                       \
      sr11 = srcT_[1792];                                                       \
      sr12 = srcG_[1024];                                                       \
      sr13 = srcG_[1280];                                                       \
      sr14 = srcG_[1536];                                                       \
      sr15 = srcG_[1792];                                                       \
    }                                                                           \
    if (DATA == DATA_BIN) {                                                     \
      _Pragma("unroll") for (int b_ = 0; b_ < NB; ++b_) cnext[b_] = codes[(size_t)((RB) + b_) * 64]; \
    }                                                                           \
  }
#define STAGE_STORE(BUF)                                                        \
  {                                                                             \
    double2* dst_ = (double2*)(lds + (BUF) * STAGE_D) + threadIdx.x;            \
    dst_[0] = sr0;                                                              \
    dst_[256] = sr1;                                                            \
    dst_[512] = sr2;                                                            \
    dst_[768] = sr3;                                                            \
    dst_[N2] = sr4;                                                             \
    dst_[N2 + 256] = sr5;                                                       \
    dst_[N2 + 512] = sr6;                                                       \
    dst_[N2 + 768] = sr7;                                                       \
    if (U8) {                                                                   \
      dst_[1024] = sr8;                                                         \
      dst_[1280] = sr9;                                                         \
      dst_[1536] = sr10;                                                        \
      dst_[1792] = sr11;                                                        \
      dst_[N2 + 1024] = sr12;                                                   \
      dst_[N2 + 1280] = sr13;                                                   \
      dst_[N2 + 1536] = sr14;                                                   \
      dst_[N2 + 1792] = sr15;                                                   \
    }                                                                           \
  }

#if NBMF_LDS_DMA
  // Experimental alternative staging: LDS-DMA (global_load_lds, 1 KiB per wave-instruction, no VGPRs).
  typedef __attribute__((address_space(3))) char lds_char;
  typedef const __attribute__((address_space(1))) char glb_char;
#define STAGE_DMA(RB, BUF)                                                                         \
  {                                                                                                \
    glb_char* gT_ = (glb_char*)(a.LT + (size_t)(RB) * BLK) + lane * 16;                           \
    glb_char* gG_ = (glb_char*)(a.LG + (size_t)(RB) * BLK) + lane * 16;                           \
    lds_char* l_ = (lds_char*)(lds + (BUF) * STAGE_D);                                             \
    constexpr int PIECES_ = N2 * 16 / 1024; /* 1 KiB pieces per image */                           \
    _Pragma("unroll") for (int u_ = 0; u_ < PIECES_ / 4; ++u_) {                                   \
      const int piece_ = wave + 4 * u_;                                                            \
      __builtin_amdgcn_global_load_lds(gT_ + piece_ * 1024, l_ + piece_ * 1024, 16, 0, 0);        \
      __builtin_amdgcn_global_load_lds(gG_ + piece_ * 1024, l_ + N2 * 16 + piece_ * 1024, 16, 0, 0); \
    }                                                                                              \
    if (DATA == DATA_BIN) {                                                                        \
      _Pragma("unroll") for (int b_ = 0; b_ < NB; ++b_) cnext[b_] = codes[(size_t)((RB) + b_) * 64]; \
    }                                                                                              \
  }
#endif

  uint32_t ccur[NB];
#if NBMF_LDS_DMA
  STAGE_DMA(rb0, 0);
#else
  STAGE_LOAD(rb0);
  STAGE_STORE(0);
#endif
#pragma unroll
  for (int b = 0; b < NB; ++b) ccur[b] = cnext[b];
  __syncthreads();

  int buf = 0;
  for (int rb = rb0; rb < rb1; rb += NB) {
    // issue next stage's global loads now; they land while this stage computes (clamped: the last
    // stage re-reads an in-range block, harmlessly)
#if NBMF_LDS_DMA
    STAGE_DMA(min(rb + NB, a.Rb - NB), buf ^ 1);   // nobody reads buf^1 between the last barrier and the next
#else
    STAGE_LOAD(min(rb + NB, a.Rb - NB));
#endif
    __builtin_amdgcn_sched_barrier(0);   // keep the loads here: hipcc otherwise sinks each one next to its ds_write
    const double* base = lds + buf * STAGE_D;

    // Operand fragments are pulled from LDS in groups of (at most) 8, one group ahead of the MFMAs that
    // use them (8 x 64 cycles of MFMA cover the LDS latency); keeps the fragment registers at 4 x 16.
    constexpr int GT = (S < 8) ? S : 8;          // Theta k-steps per group
    constexpr int NGT = S / GT;
    constexpr int PB = 4 * KB;                   // back-product operands per tile, order p = r*KB + kb
    constexpr int GB = (PB < 8) ? PB : 8;
    constexpr int NGB = PB / GB;
    double tcur[GT];
#pragma unroll
    for (int i = 0; i < GT; ++i) tcur[i] = base[i * 64 + lane];

#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double* ldsT = base + b * BLK;
      const double* ldsG = base + NB * BLK + b * BLK;

      // ---- Theta tile: rows 16(rb+b)+4r+q, column 16cb+c in register r of lane (q,c).
      // (two interleaved accumulation chains: a dependent f64 MFMA issues ~10 % slower than an
      //  independent one when the SIMD's other wave is not there to fill the gap)
      constexpr bool DUAL = NBMF_DUAL_THETA && (KB == 8 && MODE == MODE_H);   // only where one wave owns the SIMD
      d4 th = {0, 0, 0, 0}, th2 = {0, 0, 0, 0};
#pragma unroll
      for (int g = 0; g < NGT; ++g) {
        double tnxt[GT];
        if (g + 1 < NGT) {
#pragma unroll
          for (int i = 0; i < GT; ++i) tnxt[i] = ldsT[((g + 1) * GT + i) * 64 + lane];
          __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs that hide them
        }
#pragma unroll
        for (int i = 0; i < GT; i += 2) {
          th = __builtin_amdgcn_mfma_f64_16x16x4f64(tcur[i], rf[g * GT + i], th, 0, 0, 0);
          if (DUAL)
            th2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tcur[i + 1], rf[g * GT + i + 1], th2, 0, 0, 0);
          else
            th = __builtin_amdgcn_mfma_f64_16x16x4f64(tcur[i + 1], rf[g * GT + i + 1], th, 0, 0, 0);
        }
        if (g + 1 < NGT) {
#pragma unroll
          for (int i = 0; i < GT; ++i) tcur[i] = tnxt[i];
        }
      }
      if (DUAL) th += th2;
      if (MODE == MODE_L && a.clip) {
#pragma unroll
        for (int r = 0; r < 4; ++r) th[r] = fmin(fmax(th[r], 0.0), 1.0);
      }

      // first back-product operand group: issued now, lands during the ratio arithmetic
      double gcur[GB];
      if (MODE != MODE_L) {
#pragma unroll
        for (int i = 0; i < GB; ++i) gcur[i] = ldsG[((i % KB) * 4 + (i / KB)) * 64 + lane];
      } else if (b + 1 < NB) {
#pragma unroll
        for (int i = 0; i < GT; ++i) tcur[i] = base[(b + 1) * BLK + i * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);

      // ---- ratios
      double R1[4], R2[4];
      if (DATA == DATA_BIN) {
        const uint32_t code = ccur[b];
        double dd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // m = all-ones where this entry is an observed one (ym), else 0
          const uint32_t m = (uint32_t)(-(int)((code >> (8 * r)) & 1u));
          // d = ym ? Theta+eps : (1-Theta)+eps as fma(s, Theta, z) + eps with (s, z) = (1, 0) or (-1, 1):
          // one rounding for 1-Theta, exact for Theta, so bit-identical to the reference expressions
          // (including Theta > 1, which the un-normalised start of transform() can produce)
          const double z = mk_double(0u, ~m & 0x3FF00000u);
          const double sg = mk_double(0u, (~m & 0x80000000u) | 0x3FF00000u);
          const double d = __builtin_fma(sg, th[r], z) + eps;
          if (MODE == MODE_L) {
            // strict: only observed entries (ones or zeros) enter the product; otherwise every entry
            // does, exactly as in MODE_H (pad entries are divided out later)
            const uint32_t keep = a.strict ? (m | (uint32_t)(-(int)((code >> (8 * r + 1)) & 1u))) : 0xFFFFFFFFu;
            dd[r] = mk_double((uint32_t)__double2loint(d) & keep,
                              ((uint32_t)__double2hiint(d) & keep) | (~keep & 0x3FF00000u));   // keep ? d : 1.0
            continue;
          }
          const double rr = rcp_nr(d);
          dd[r] = d;
          const uint32_t rlo = (uint32_t)__double2loint(rr), rhi = (uint32_t)__double2hiint(rr);
          if (MODE == MODE_H) {
            // Every entry that is not an observed one acts as an observed zero here;
            // pad entries have Theta == 0 exactly and their operand rows/columns are zero, so they only
            // touch the loss product, by the constant (1+eps) that is divided out before the loss.
            R1[r] = mk_double(rlo & m, rhi & m);        // ym ? rr : 0
            R2[r] = mk_double(rlo & ~m, rhi & ~m);      // ym ? 0 : rr
          } else {
            const uint32_t zo = (uint32_t)(-(int)((code >> (8 * r + 1)) & 1u));   // observed zero
            const uint32_t keep = m | zo;
            R1[r] = mk_double(rlo & keep, (rhi ^ (zo & 0x80000000u)) & keep);     // S1 - S2: +rr, -rr or 0
            s2 += mk_double(rlo & zo, rhi & zo);
          }
        }
        if (MODE == MODE_H || MODE == MODE_L) {
          prod *= (dd[0] * dd[1]) * (dd[2] * dd[3]);   // >= 1e-32 per tile: no underflow before frexp
          int e;
          prod = frexp(prod, &e);
          pexp += e;
        }
      } else {
        const d4 y4 = yv[(size_t)(rb + b) * 64];
        d4 m4 = {1, 1, 1, 1};
        if (DATA == DATA_F64M) m4 = mv[(size_t)(rb + b) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double y = y4[r];
          const bool valid = y >= 0.0;    // pad entries are stored as -1
          const double t = th[r];
          const double t1 = t + eps;
          const double t2 = (1.0 - t) + eps;
          if (MODE == MODE_L) {
            if (a.strict) {
              // mask * (Y log(Theta+eps) + (1-Y) log(1-Theta+eps)): examples/reproduce_magron2022.py:40-47
              const double wgt = (DATA == DATA_F64M) ? m4[r] : 1.0;
              llsum += (valid && wgt != 0.0) ? wgt * (y * log_fast(t1) + (1.0 - y) * log_fast(t2)) : 0.0;
            } else {
              const double ym = (DATA == DATA_F64M) ? y * m4[r] : y;
              llsum += valid ? (ym * log_fast(t1) + (1.0 - ym) * log_fast(t2)) : 0.0;   // as MODE_H, :150,154
            }
          } else if (MODE == MODE_H) {
            const double ym = (DATA == DATA_F64M) ? y * m4[r] : y;    // Y*mask, _solver.py:30
            R1[r] = valid ? ym / t1 : 0.0;                            // :42
            R2[r] = valid ? (1.0 - ym) / t2 : 0.0;                    // :43 (1 - Y*mask)
            llsum += valid ? (ym * log_fast(t1) + (1.0 - ym) * log_fast(t2)) : 0.0;   // :150,154
          } else {
            const double yo = (DATA == DATA_F64M) ? y * m4[r] : y;                 // Y.T*mask.T, :31
            const double zo = (DATA == DATA_F64M) ? (1.0 - y) * m4[r] : (1.0 - y);  // (1-Y).T*mask.T, :32
            const double s1v = valid ? yo / t1 : 0.0;
            const double s2v = valid ? zo / t2 : 0.0;
            R1[r] = s1v - s2v;
            s2 += s2v;
          }
        }
      }

      // ---- back-products: accumulator registers of Theta are the B operands (rows 4r..4r+3)
      if (MODE != MODE_L) {
#pragma unroll
      for (int g = 0; g < NGB; ++g) {
        double gnxt[GB];
        if (g + 1 < NGB) {
#pragma unroll
          for (int i = 0; i < GB; ++i) {
            const int p = (g + 1) * GB + i;
            gnxt[i] = ldsG[((p % KB) * 4 + (p / KB)) * 64 + lane];
          }
          __builtin_amdgcn_sched_barrier(0);
        } else if (b + 1 < NB) {
          // last group: fetch the first Theta group of the next tile of this stage
#pragma unroll
          for (int i = 0; i < GT; ++i) tcur[i] = base[(b + 1) * BLK + i * 64 + lane];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
          const int p = g * GB + i, r = p / KB, kb = p % KB;
          acc1[kb] = __builtin_amdgcn_mfma_f64_16x16x4f64(gcur[i], R1[r], acc1[kb], 0, 0, 0);
          if (MODE == MODE_H) acc2[kb] = __builtin_amdgcn_mfma_f64_16x16x4f64(gcur[i], R2[r], acc2[kb], 0, 0, 0);
        }
        if (g + 1 < NGB) {
#pragma unroll
          for (int i = 0; i < GB; ++i) gcur[i] = gnxt[i];
        }
      }
      }
    }

    // write the prefetched stage into the other buffer; its last readers finished before the previous
    // barrier, and this stage's readers use `buf`
#if !NBMF_LDS_DMA
    STAGE_STORE(buf ^ 1);
#endif
#pragma unroll
    for (int b = 0; b < NB; ++b) ccur[b] = cnext[b];
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue: slabs [chunk][k][column]
  const int q = lane >> 4, c = lane & 15;
  if (MODE == MODE_W) {
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
  }
  double* o1 = a.out1 + (size_t)chunk * K * a.C_alloc + (size_t)cb * 16 + c;
  double* o2 = (MODE == MODE_H) ? a.out2 + (size_t)chunk * K * a.C_alloc + (size_t)cb * 16 + c : nullptr;
  if (MODE != MODE_L) {
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t k = 16 * kb + 4 * r + q;
      if (MODE == MODE_H) {
        o1[k * a.C_alloc] = acc1[kb][r];
        o2[k * a.C_alloc] = acc2[kb][r];
      } else {
        o1[k * a.C_alloc] = acc1[kb][r] + s2;
      }
    }
  }
  }
#undef STAGE_LOAD
#undef STAGE_STORE
  if (MODE == MODE_H || MODE == MODE_L) {
    double ll;
    if (DATA == DATA_BIN)
      ll = log(prod) + (double)pexp * 0.6931471805599453094;
    else
      ll = llsum;
    ll = wave_sum(ll);
    // one partial per workgroup: the four waves in wave order (fixed -> reproducible)
    __syncthreads();                       // all waves are done with the staging buffers
    if (lane == 0) lds[wave] = ll;
    __syncthreads();
    if (threadIdx.x == 0)
      a.lossbuf[(size_t)chunk * gridDim.x + blockIdx.x] = ((lds[0] + lds[1]) + lds[2]) + lds[3];
  }
}

// ------------------------------------------------------------------------------------------
// Deterministic block-wide sum (256 threads): thread t adds p[t*stride], p[(t+256)*stride], ... in
// order, then a fixed butterfly inside each wave and the four waves in order.  Every thread must call
// it; the result is returned to thread 0 (other threads get garbage).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double ordered_sum256(const double* __restrict__ p, int n, int stride, double* sh4) {
  double s = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) s += p[(size_t)j * stride];
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = s;
  __syncthreads();
  return ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
}

// Multi-GPU only: ordered reduction of the H-pass slabs into the all-reduce payload
// Pbuf = [P1 | P2 | loglik]: Pbuf[t][k][j] = sum_chunks slab[t][chunk][k][j].
__global__ __launch_bounds__(256) void reduce_h_kernel(const double* __restrict__ slab1, const double* __restrict__ slab2,
                                                       const double* __restrict__ lossbuf, double* __restrict__ Pbuf,
                                                       int chunks, long long per /* K_pad * nA */, int n_loss,
                                                       double ll_pad, const int* done) {
  __shared__ double sh4[4];
  if (*done) return;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < per) {
    double s1 = 0, s2 = 0;
    for (int ch = 0; ch < chunks; ++ch) {
      s1 += slab1[(size_t)ch * per + i];
      s2 += slab2[(size_t)ch * per + i];
    }
    Pbuf[i] = s1;
    Pbuf[per + i] = s2;
  }
  if (blockIdx.x == 0) {
    const double s = ordered_sum256(lossbuf, n_loss, 1, sh4);
    if (threadIdx.x == 0) Pbuf[2 * per] = s - ll_pad;   // binary path: pad entries each contributed log(1+eps)
  }
}

// Column-sharded runs: ordered sums of the Beta log-prior partials into the scalar exchange slot.
__global__ __launch_bounds__(256) void prior_reduce_kernel(const double* __restrict__ prior, int n_prior,
                                                           double* __restrict__ out2, const int* done) {
  __shared__ double sh4[4];
  if (*done) return;
  const double a = ordered_sum256(prior, n_prior, 2, sh4);
  const double b = ordered_sum256(prior + 1, n_prior, 2, sh4);
  if (threadIdx.x == 0) {
    out2[0] = a;
    out2[1] = b;
  }
}

// Column-sharded runs: ordered sum of the W-pass slabs into the all-reduce payload Qbuf[k][i].
__global__ __launch_bounds__(256) void reduce_w_kernel(const double* __restrict__ slab, double* __restrict__ Qbuf,
                                                       int chunks, long long per, const int* done) {
  if (*done) return;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per) return;
  double s = 0;
  for (int ch = 0; ch < chunks; ++ch) s += slab[(size_t)ch * per + i];
  Qbuf[i] = s;
}

// Sharded evaluation sweeps: ordered sum of the log-likelihood partials into the all-reduce slot.
__global__ __launch_bounds__(256) void ll_reduce_kernel(const double* __restrict__ lossbuf, int n_loss, double ll_pad,
                                                        double* __restrict__ out, const int* done) {
  __shared__ double sh4[4];
  if (*done) return;
  const double s = ordered_sum256(lossbuf, n_loss, 1, sh4);
  if (threadIdx.x == 0) *out = s - ll_pad;
}

// Beta log-prior sums of H (natural layout), per-block partials -> prior[blk][2] (_solver.py:158-159).
__device__ __forceinline__ void block_sum2(double a, double b, double* out2) {
  __shared__ double sh[2][8];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[0][w] = a;
    sh[1][w] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0, y = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
      x += sh[0][i];
      y += sh[1][i];
    }
    out2[0] = x;
    out2[1] = y;
  }
}

// H-update (_solver.py:42-47); P1/P2 are summed here over `chunks` partial slabs in chunk order
// (single GPU: the H-pass slabs; multi-GPU: the all-reduced Pbuf, chunks = 1).  Writes natural, T and
// G forms.  One thread per (k, j); j fastest.
__global__ __launch_bounds__(256) void h_update_kernel(const double* __restrict__ src1, const double* __restrict__ src2,
                                                       int chunks, double* __restrict__ Hn, double* __restrict__ HT,
                                                       double* __restrict__ HG, double* __restrict__ prior, int K, int KP,
                                                       long long n, long long nA, double am1, double bm1, double eps,
                                                       const int* done) {
  if (*done) return;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long k = idx / nA, j = idx % nA;
  const size_t per = (size_t)KP * nA;
  double la = 0, lb = 0;
  if (k < KP) {
    double h = 0.0;
    if (k < K && j < n) {
      double p1 = 0, p2 = 0;
      for (int ch = 0; ch < chunks; ++ch) {
        p1 += src1[(size_t)ch * per + idx];
        p2 += src2[(size_t)ch * per + idx];
      }
      const double hold = Hn[idx];
      const double num = hold * p1 + am1;             // :42
      const double den = (1.0 - hold) * p2 + bm1;     // :43
      h = num / (num + den + eps);                    // :46
      h = (h < eps) ? eps : ((h > 1.0 - eps) ? 1.0 - eps : h);   // np.clip, :47 (a NaN stays a NaN, as in NumPy)
      la = log(h + eps);                              // :158
      lb = log(1.0 - h + eps);                        // :159
    }
    Hn[idx] = h;
    const long long jb = j >> 4, c = j & 15;
    HT[(jb * KP + k) * 16 + c] = h;
    HG[jb * KP * 16 + (k >> 4) * 256 + c * 16 + (k & 15)] = h;
  }
  block_sum2(la, lb, prior + 2 * (size_t)blockIdx.x);
}

// Prior sums only (used after nbmf_set_factors).
__global__ __launch_bounds__(256) void prior_kernel(const double* __restrict__ Hn, double* __restrict__ prior, int K,
                                                    int KP, long long n, long long nA, double eps) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long k = idx / nA, j = idx % nA;
  double la = 0, lb = 0;
  if (k < K && j < n) {
    const double h = Hn[idx];
    la = log(h + eps);
    lb = log(1.0 - h + eps);
  }
  block_sum2(la, lb, prior + 2 * (size_t)blockIdx.x);
}

// Loss assembly + stop rule (_solver.py:158-175), one 256-thread block.
//   ll = ordered sum of ll_src[0..n_ll) - ll_pad   (single GPU: the per-wave H-pass partials;
//                                                     multi-GPU: the all-reduced Pbuf tail, n_ll = 1)
//   scal[0] = previous loss, flags[0] = done, flags[1] = n_iter
__global__ __launch_bounds__(256) void finalize_kernel(const double* __restrict__ ll_src, int n_ll, double ll_pad,
                                                       const double* __restrict__ prior, int n_prior, double am1,
                                                       double bm1, double n_obs, double* __restrict__ losses, int t,
                                                       double tol, double* __restrict__ scal, int* __restrict__ flags) {
  __shared__ double sh[3][4];
  if (flags[0]) return;
  if (t < 0) {
    // replayed (hipGraph) form: the loss index lives on the device; the very first call of a run only
    // arms the counter (the H-pass of iteration 0 has no finished iteration to score)
    const int armed = flags[3];
    __syncthreads();
    if (!armed) {
      if (threadIdx.x == 0) flags[3] = 1;
      return;
    }
    t = flags[2];
  }
  // three ordered sums in one sweep (same per-thread / butterfly / wave order as ordered_sum256, so the
  // single-GPU and the all-reduced paths agree bit for bit); the loads of the three streams overlap
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const int n_max = n_ll > n_prior ? n_ll : n_prior;
#pragma unroll 4
  for (int j = threadIdx.x; j < n_max; j += 256) {
    if (j < n_ll) s0 += ll_src[j];
    if (j < n_prior) {
      s1 += prior[2 * (size_t)j];
      s2 += prior[2 * (size_t)j + 1];
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s0;
    sh[1][threadIdx.x >> 6] = s1;
    sh[2][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  const double ll = (((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3]) - ll_pad;
  const double sa = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
  const double sb = ((sh[2][0] + sh[2][1]) + sh[2][2]) + sh[2][3];
  if (threadIdx.x == 0) {
    const double A = am1 * sa;
    const double B = bm1 * sb;
    const double loss = -(ll + A + B) / n_obs;   // :162
    losses[t] = loss;
    flags[1] = t + 1;
    flags[2] = t + 1;
    if (t > 0) {
      const double prev = scal[0];
      if (fabs(prev - loss) / fabs(prev) < tol) flags[0] = 1;   // :169-174
    }
    scal[0] = loss;
  }
}

// W-update (_solver.py:53-57).  Block = 32 columns x 8 k-groups: every thread reduces the W-pass slabs
// for its (k, column) entries (coalesced over columns) into an LDS tile of products W*Q/div; one
// thread per column then sums the K products in k order (the order of numpy's sum(axis=0)) and
//   projection 0: divides by the column sum (:57);
//   projection 1 (extension, README.md:27-35): div = per-row observed count, then Euclidean
//                 projection onto the simplex by Michelot's active-set iteration (unique minimiser);
// all threads write natural, T and G forms.
constexpr int WU_COLS = 32, WU_GROUPS = 8;
__global__ __launch_bounds__(256) void w_update_kernel(const double* __restrict__ slab, int chunks, double* __restrict__ Wn,
                                                       double* __restrict__ WT, double* __restrict__ WG, int K, int KP,
                                                       long long m, long long mA, double n_div,
                                                       const double* __restrict__ rowcnt, int projection,
                                                       const int* done) {
  extern __shared__ __attribute__((aligned(16))) double tile[];   // [KP][WU_COLS] products, then [2][WU_COLS] scale/tau
  if (*done) return;
  const int c = threadIdx.x & (WU_COLS - 1), g = threadIdx.x / WU_COLS;
  const long long i = (long long)blockIdx.x * WU_COLS + c;
  const size_t per = (size_t)KP * mA;
  const bool live = i < m;
  double div = n_div;
  if (live && projection == NBMF_PROJ_DUCHI && rowcnt) div = fmax(rowcnt[i], 1.0);
  for (int k = g; k < K; k += WU_GROUPS) {
    double w = 0.0;
    if (live) {
      double qv = 0.0;
      for (int ch = 0; ch < chunks; ++ch) qv += slab[(size_t)ch * per + (size_t)k * mA + i];
      w = (Wn[(size_t)k * mA + i] * qv) / div;   // :53-54
    }
    tile[k * WU_COLS + c] = w;
  }
  __syncthreads();
  double* par = tile + (size_t)KP * WU_COLS;      // [0][c] = scale, [1][c] = tau
  if (g == 0) {
    double sum = 0.0;
    for (int k = 0; k < K; ++k) sum += tile[k * WU_COLS + c];
    double tau = 0.0, scale = 1.0;
    if (projection == NBMF_PROJ_DUCHI) {
      tau = (sum - 1.0) / K;
      int cnt = K;
      for (int it = 0; it < K; ++it) {
        double s = 0.0;
        int c2 = 0;
        for (int k = 0; k < K; ++k) {
          const double v = tile[k * WU_COLS + c];
          if (v > tau) {
            s += v;
            ++c2;
          }
        }
        if (c2 == 0) break;
        const bool same = (c2 == cnt);
        tau = (s - 1.0) / c2;
        cnt = c2;
        if (same) break;
      }
    } else {
      scale = sum;
    }
    par[c] = scale;
    par[WU_COLS + c] = tau;
  }
  __syncthreads();
  const double scale = par[c], tau = par[WU_COLS + c];
  const long long ib = i >> 4, cc = i & 15;
  for (int k = g; k < KP; k += WU_GROUPS) {
    double w = 0.0;
    if (live && k < K) {
      const double v = tile[k * WU_COLS + c];
      w = (projection == NBMF_PROJ_DUCHI) ? fmax(v - tau, 0.0) : v / scale;   // :57
    }
    Wn[(size_t)k * mA + i] = w;
    WT[(ib * KP + k) * 16 + cc] = w;
    WG[ib * KP * 16 + (k >> 4) * 256 + cc * 16 + (k & 15)] = w;
  }
}

// Factor upload: natural true-size [K][len] (staging) -> padded natural, T, G.
__global__ void set_factor_kernel(const double* __restrict__ src, double* __restrict__ Fn, double* __restrict__ FT,
                                  double* __restrict__ FG, int K, int KP, long long len, long long lenA) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)KP * lenA) return;
  const long long k = idx / lenA, x = idx % lenA;
  const double v = (k < K && x < len) ? src[k * len + x] : 0.0;
  Fn[idx] = v;
  const long long xb = x >> 4, c = x & 15;
  FT[(xb * KP + k) * 16 + c] = v;
  FG[xb * KP * 16 + (k >> 4) * 256 + c * 16 + (k & 15)] = v;
}

__global__ void get_factor_kernel(const double* __restrict__ Fn, double* __restrict__ dst, int K, long long len,
                                  long long lenA) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)K * len) return;
  const long long k = idx / len, x = idx % len;
  dst[idx] = Fn[k * lenA + x];
}

// ------------------------------------------------------------------------------------------
// Pack kernel: raw row-major chunk of the user's matrix -> the two tile-ordered images.
// One wave per 16x16 tile of the user's matrix X (rows u, columns v).
//   transposed = 0: Y = X   (i = u, j = v);   transposed = 1: Y = X^T (i = v, j = u).
//   image A (H-pass): [strip over j][block over i][lane (q,c)][r] = Y[16ib+4r+q][16jb+c]
//   image B (W-pass): [strip over i][block over j][lane (q,c)][r] = Y[16ib+c][16jb+4r+q]
// stats[0] = count of observed (mask != 0) in-range entries, stats[1] = #entries outside [0,1] or
// non-finite, stats[2] = #non-binary data entries, stats[3] = #non-binary mask entries.
// ------------------------------------------------------------------------------------------
struct PackArgs {
  const double* x;        // chunk base: rows [u0, u0+urows) of X
  const void* mask;       // same chunk of the mask or nullptr
  int mask_kind;
  long long ldx, ldmask;  // elements
  long long u0, urows;    // chunk row range in X
  long long U, V;         // X dims
  int transposed;
  int binary;             // 1: write byte codes, 0: write doubles (+ mask doubles)
  void *dataA, *dataB, *maskA, *maskB;
  long long RbA, RbB;     // row blocks of image A (= mA/16) and of image B (= nA/16)
  unsigned long long* stats;
  double* rowcnt;         // unused here (filled by rowcount kernel)
};

__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
  __shared__ double tv[4][16][17];
  __shared__ double tm[4][16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const long long vb = (long long)blockIdx.x * 4 + wave;            // tile column in X
  const long long ub = a.u0 / 16 + blockIdx.y;                      // tile row in X
  unsigned long long n_obs = 0, n_bad = 0, n_nonbin = 0, n_mnonbin = 0;
  // load style "A": element (row 4r+q, col c) of the X tile
  double va[4], ma[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long u = ub * 16 + 4 * r + q, v = vb * 16 + c;
    double x = -1.0, mk = 0.0;
    if (u < a.U && u < a.u0 + a.urows && v < a.V) {
      x = a.x[(u - a.u0) * a.ldx + v];
      mk = 1.0;
      if (a.mask_kind == NBMF_MASK_F64) mk = ((const double*)a.mask)[(u - a.u0) * a.ldmask + v];
      if (a.mask_kind == NBMF_MASK_U8) mk = ((const unsigned char*)a.mask)[(u - a.u0) * a.ldmask + v] ? 1.0 : 0.0;
      if (!(x >= 0.0 && x <= 1.0)) ++n_bad;
      if (x != 0.0 && x != 1.0) ++n_nonbin;
      if (mk != 0.0 && mk != 1.0) ++n_mnonbin;
      if (mk != 0.0) ++n_obs;
    }
    va[r] = x;
    ma[r] = mk;
    tv[wave][4 * r + q][c] = x;
    tm[wave][4 * r + q][c] = mk;
  }
  __syncthreads();
  // style "B": element (row c, col 4r+q)
  double vbv[4], mbv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    vbv[r] = tv[wave][c][4 * r + q];
    mbv[r] = tm[wave][c][4 * r + q];
  }
  // which style feeds which image
  const double* forA_v = a.transposed ? vbv : va;
  const double* forA_m = a.transposed ? mbv : ma;
  const double* forB_v = a.transposed ? va : vbv;
  const double* forB_m = a.transposed ? ma : mbv;
  // tile coordinates in Y: ib (block over i), jb (block over j)
  const long long ib = a.transposed ? vb : ub;
  const long long jb = a.transposed ? ub : vb;
  if (ib < a.RbA && jb < a.RbB) {
    const size_t ia = ((size_t)jb * a.RbA + ib) * 64 + lane;   // image A: strip jb, block ib
    const size_t ibx = ((size_t)ib * a.RbB + jb) * 64 + lane;  // image B: strip ib, block jb
    if (a.binary) {
      uint32_t ca = 0, cbb = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        {
          const double x = forA_v[r], mk = forA_m[r];
          uint32_t code = 0;
          if (x >= 0.0) code = CB_VALID | ((x != 0.0 && mk != 0.0) ? CB_YM : 0u) | ((x == 0.0 && mk != 0.0) ? CB_ZOBS : 0u);
          ca |= code << (8 * r);
        }
        {
          const double x = forB_v[r], mk = forB_m[r];
          uint32_t code = 0;
          if (x >= 0.0) code = CB_VALID | ((x != 0.0 && mk != 0.0) ? CB_YM : 0u) | ((x == 0.0 && mk != 0.0) ? CB_ZOBS : 0u);
          cbb |= code << (8 * r);
        }
      }
      ((uint32_t*)a.dataA)[ia] = ca;
      ((uint32_t*)a.dataB)[ibx] = cbb;
    } else {
      ((d4*)a.dataA)[ia] = d4{forA_v[0], forA_v[1], forA_v[2], forA_v[3]};
      ((d4*)a.dataB)[ibx] = d4{forB_v[0], forB_v[1], forB_v[2], forB_v[3]};
      if (a.mask_kind != NBMF_MASK_NONE) {
        ((d4*)a.maskA)[ia] = d4{forA_m[0], forA_m[1], forA_m[2], forA_m[3]};
        ((d4*)a.maskB)[ibx] = d4{forB_m[0], forB_m[1], forB_m[2], forB_m[3]};
      }
    }
  }
  // integer statistics (order-independent)
  for (int off = 32; off >= 1; off >>= 1) {
    n_obs += __shfl_xor(n_obs, off, 64);
    n_bad += __shfl_xor(n_bad, off, 64);
    n_nonbin += __shfl_xor(n_nonbin, off, 64);
    n_mnonbin += __shfl_xor(n_mnonbin, off, 64);
  }
  if (lane == 0) {
    if (n_obs) atomicAdd(&a.stats[0], n_obs);
    if (n_bad) atomicAdd(&a.stats[1], n_bad);
    if (n_nonbin) atomicAdd(&a.stats[2], n_nonbin);
    if (n_mnonbin) atomicAdd(&a.stats[3], n_mnonbin);
  }
}

// Per-internal-row observed weight (Duchi extension, README.md:32-35): thread per internal row i,
// walking image B (strip ib, blocks over j) in order -> deterministic.
__global__ void rowcount_kernel(const void* dataB, const void* maskB, int data_kind, long long RbB, long long m,
                                double* rowcnt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const long long ib = i >> 4;
  const int c = i & 15;
  double s = 0.0;
  for (long long jb = 0; jb < RbB; ++jb) {
    for (int qq = 0; qq < 4; ++qq) {
      const size_t base = ((size_t)ib * RbB + jb) * 64 + qq * 16 + c;
      if (data_kind == DATA_BIN) {
        const uint32_t code = ((const uint32_t*)dataB)[base];
        for (int r = 0; r < 4; ++r) s += ((code >> (8 * r)) & (CB_YM | CB_ZOBS)) ? 1.0 : 0.0;
      } else {
        const d4 y = ((const d4*)dataB)[base];
        d4 mk = {1, 1, 1, 1};
        if (data_kind == DATA_F64M) mk = ((const d4*)maskB)[base];
        for (int r = 0; r < 4; ++r) s += (y[r] >= 0.0) ? mk[r] : 0.0;
      }
    }
  }
  rowcnt[i] = s;
}

__global__ void unary_test_kernel(int op, const double* __restrict__ x, double* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (op == 0) ? rcp_nr(x[i]) : log_fast(x[i]);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct nbmf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t m = 0, n = 0, mA = 0, nA = 0;
  int k = 0, KP = 0, KB = 0;
  int data_kind = -1;   // -1 = nothing uploaded
  void *dataA = nullptr, *dataB = nullptr, *maskA = nullptr, *maskB = nullptr;
  double n_obs = 0, n_obs_global = 0;
  double* rowcnt = nullptr;
  double *Wn = nullptr, *WT = nullptr, *WG = nullptr, *Hn = nullptr, *HT = nullptr, *HG = nullptr;
  bool have_factors = false;
  int chunksH = 0, CH_H = 0, chunksW = 0, CH_W = 0;
  int *cstartH = nullptr, *cstartW = nullptr;   // device: chunk boundaries of the two sweeps
  double *slabH = nullptr, *slabW = nullptr, *Pbuf = nullptr, *lossbuf = nullptr, *prior = nullptr, *scal = nullptr;
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
  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<int> ev_kind;     // 0 = H-pass, 1 = W-pass
  size_t ev_used = 0;
  double t_ms[2] = {0, 0};
  int t_n[2] = {0, 0};
};

namespace {

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
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return NBMF_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (const char* nm : names) {
    lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(NBMF_ERR_COMM, "cannot load librccl: %s", dlerror());
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, Uid, int))dlsym(lib, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
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

// ---- pass launch ----------------------------------------------------------------------------
template <int KB, int DATA, int MODE>
hipError_t launch_pass_t(const PassArgs& a, int chunks, hipStream_t st) {
  dim3 grid(a.Cb / WG_WAVES, chunks);
  constexpr int lds_bytes = (NBMF_STAGE_HALF && KB <= 4) ? STAGE_BYTES : 2 * STAGE_BYTES;
  if (lds_bytes > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)pass_kernel<KB, DATA, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((pass_kernel<KB, DATA, MODE>), grid, dim3(256), lds_bytes, st, a);
  return hipGetLastError();
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
  const int lds_bytes = (NBMF_STAGE_HALF && KB <= 4) ? STAGE_BYTES : 2 * STAGE_BYTES;
  if (!f || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 256, lds_bytes) != hipSuccess || n < 1) n = 2;
  return std::min(n, 8);
}

// Split a sweep of Rb row blocks into chunks: aim for ~2048 workgroups (several rounds over the 512-768
// resident ones, so the dispatcher can even out slow workgroups) without making the sweeps shorter than
// 64 blocks (each workgroup pays a fixed prologue/epilogue and writes its own partial slab).
// Measured at c3 / K=64: 768 workgroups = exactly one round of the 768 resident slots is SLOWER (3.23 ms)
// than 2048 (3.15 ms); anything from 1536 to 4608 is within 1 %.  `slots` is kept for diagnostics.
// NBMF_TARGET_WGS=<n> overrides the target (tuning experiments only).
void pick_chunks(int strips_groups, int Rb, int NB, int slots, int* chunks, int* CH) {
  (void)slots;
  int target = 2048;
  if (const char* e = getenv("NBMF_TARGET_WGS")) target = std::max(1, atoi(e));
  int want = (target + strips_groups - 1) / strips_groups;
  const int max_chunks = std::max(1, Rb / NB);
  want = std::min(std::max(want, 1), max_chunks);
  int ch = (Rb + want - 1) / want;
  ch = std::max(ch, std::min(Rb, 64));
  ch = (int)round_up(ch, NB);
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
  EvScope(nbmf_ctx* c_, int kind_) : c(c_), kind(kind_) {
    if (!c->timing) return;
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
    hipEventRecord(c->ev[slot], c->stream);
  }
  ~EvScope() {
    if (slot != (size_t)-1) hipEventRecord(c->ev[slot + 1], c->stream);
  }
};

void timing_collect(nbmf_ctx* c) {
  // caller has synchronised the stream
  for (size_t s = 0; s + 1 < c->ev_used; s += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev[s], c->ev[s + 1]) == hipSuccess) {
      c->t_ms[c->ev_kind[s / 2]] += ms;
      c->t_n[c->ev_kind[s / 2]] += 1;
    }
  }
  c->ev_used = 0;
}

// In-place sum over ranks of `count` doubles at device pointer `p`, on the context's stream.
int all_reduce_inplace(nbmf_ctx* c, double* p, size_t count) {
  if (c->comm) {
    NCCLCHK(g_rccl.AllReduce(p, p, count, kNcclFloat64, kNcclSum, c->comm, c->stream));
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

inline bool is_sharded(const nbmf_ctx* c) { return c->comm || c->host_reduce; }

// Binary path: the H-pass multiplies every entry of the padded mA x nA grid into the likelihood
// product; a pad entry has Theta == 0 and is not an observed one, so it contributes exactly
// fl(fl(1-0)+eps) = 1+eps.  Their total is removed before the loss is assembled (and before any
// all-reduce).
double ll_pad_of(const nbmf_ctx* c) {
  const double n_pad = (double)c->mA * (double)c->nA - (double)c->m * (double)c->n;
  return (c->data_kind == DATA_BIN) ? n_pad * log(1.0 + c->eps) : 0.0;
}

// What travels after a sweep over image A (H-pass or Theta-only sweep), and where the global
// log-likelihood ends up (c->ll_ptr):
//   single GPU        nothing; finalize sums the per-wave partials itself
//   axis 0 (rows)     H-pass: [P1 | P2 | loglik] in Pbuf; Theta-only sweep: the loglik scalar
//   axis 1 (columns)  [loglik, prior A, prior B]: the products stay local, the scalars do not
int enqueue_exchange_after_sweep(nbmf_ctx* c, const PassArgs& a, bool with_products, int strict) {
  const int n_loss = c->chunksH * (a.Cb / WG_WAVES);   // one log-likelihood partial per workgroup
  const double pad = strict ? 0.0 : ll_pad_of(c);
  if (!is_sharded(c)) {
    c->ll_ptr = nullptr;
    return NBMF_OK;
  }
  const long long per = (long long)c->KP * c->nA;
  if (c->shard_axis == 0 && with_products) {
    hipLaunchKernelGGL(reduce_h_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, c->stream, a.out1, a.out2,
                       c->lossbuf, c->Pbuf, c->chunksH, per, n_loss, pad, c->flags);
    HIPCHK(hipGetLastError());
    if (int rc = all_reduce_inplace(c, c->Pbuf, (size_t)(2 * per + 1))) return rc;
    c->ll_ptr = c->Pbuf + 2 * per;
    return NBMF_OK;
  }
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

int enqueue_h_pass(nbmf_ctx* c) {
  PassArgs a{};
  a.data = c->dataA;
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
  {
    EvScope ev(c, 0);
    HIPCHK(launch_pass<MODE_H>(c->KB, c->data_kind, a, c->chunksH, c->stream));
  }
  if (int rc = enqueue_exchange_after_sweep(c, a, /*with_products=*/true, /*strict=*/0)) return rc;
  return NBMF_OK;
}

// Theta-only sweep (no back-products): the log-likelihood of the current factors at a third of the
// H-pass's MFMA work; the per-wave partials land in lossbuf exactly as an H-pass leaves them.
int enqueue_loglik_pass(nbmf_ctx* c, int strict, int clip = 0) {
  PassArgs a{};
  a.data = c->dataA;
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
  a.strict = strict;
  a.clip = clip;
  HIPCHK(launch_pass<MODE_L>(c->KB, c->data_kind, a, c->chunksH, c->stream));
  if (int rc = enqueue_exchange_after_sweep(c, a, /*with_products=*/false, strict)) return rc;
  return NBMF_OK;
}

int enqueue_finalize(nbmf_ctx* c, int t, double tol, bool loglik_only = false, int strict = 0) {
  // single GPU: per-wave partials + pad correction here; sharded: the exchanged scalar (pad already removed)
  const bool sh = is_sharded(c) && c->ll_ptr;
  const double* ll_src = sh ? c->ll_ptr : c->lossbuf;
  const int n_ll = sh ? 1 : c->chunksH * (int)(c->nA / 16 / WG_WAVES);
  const double pad = (sh || strict) ? 0.0 : ll_pad_of(c);
  // axis 1: the prior sums were exchanged with the log-likelihood
  const bool prior_x = sh && c->shard_axis == 1;
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, c->stream, ll_src, n_ll, pad,
                     prior_x ? (const double*)(c->sbuf + 1) : (const double*)c->prior,
                     loglik_only ? 0 : (prior_x ? 1 : c->n_prior_blocks),   // no prior term (and no 0 x NaN) in a pure log-likelihood
                     loglik_only ? 0.0 : c->alpha - 1.0, loglik_only ? 0.0 : c->beta - 1.0,
                     loglik_only ? -1.0 : c->n_obs_global, c->losses_d, t, tol, c->scal, c->flags);
  HIPCHK(hipGetLastError());
  return NBMF_OK;
}

int enqueue_h_update(nbmf_ctx* c) {
  const size_t per = (size_t)c->KP * c->nA;
  const bool reduced = is_sharded(c) && c->shard_axis == 0;     // products arrive all-reduced in Pbuf
  const double* s1 = reduced ? c->Pbuf : c->slabH;
  const double* s2 = reduced ? c->Pbuf + per : c->slabH + (size_t)c->chunksH * per;
  hipLaunchKernelGGL(h_update_kernel, dim3(c->n_prior_blocks), dim3(256), 0, c->stream, s1, s2,
                     reduced ? 1 : c->chunksH, c->Hn, c->HT, c->HG, c->prior, c->k, c->KP, (long long)c->n,
                     (long long)c->nA, c->alpha - 1.0, c->beta - 1.0, c->eps, c->flags);
  HIPCHK(hipGetLastError());
  return NBMF_OK;
}

int enqueue_w_step(nbmf_ctx* c, int projection) {
  PassArgs a{};
  a.data = c->dataB;
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
  {
    EvScope ev(c, 1);
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
  hipLaunchKernelGGL(w_update_kernel, dim3((unsigned)(c->mA / WU_COLS)), dim3(WU_COLS * WU_GROUPS),
                     sizeof(double) * ((size_t)c->KP + 2) * WU_COLS, c->stream, q, chunks, c->Wn, c->WT, c->WG, c->k, c->KP,
                     (long long)c->m, (long long)c->mA, n_div, c->rowcnt, projection, c->flags);
  HIPCHK(hipGetLastError());
  return NBMF_OK;
}

int ensure_losses(nbmf_ctx* c, int cap) {
  if (cap <= c->losses_cap) return NBMF_OK;
  if (c->losses_d) HIPCHK(hipFree(c->losses_d));
  c->losses_d = nullptr;
  HIPCHK(hipMalloc(&c->losses_d, sizeof(double) * (size_t)cap));
  c->losses_cap = cap;
  return NBMF_OK;
}

// Chunking of the two sweeps and the slabs that go with it; needs the storage path (it decides the
// kernels' residency), so it runs at the end of nbmf_upload.
int setup_workspaces(nbmf_ctx* c) {
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, c->device));
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const int NB = 8 / c->KB;
  const int slotsH = cus * resident_per_cu<MODE_H>(c->KB, c->data_kind);
  const int slotsW = cus * resident_per_cu<MODE_W>(c->KB, c->data_kind);
  pick_chunks((int)(c->nA / 16 / WG_WAVES), (int)(c->mA / 16), NB, slotsH, &c->chunksH, &c->CH_H);
  pick_chunks((int)(c->mA / 16 / WG_WAVES), (int)(c->nA / 16), NB, slotsW, &c->chunksW, &c->CH_W);
  const std::vector<int> bH = chunk_boundaries((int)(c->mA / 16), c->CH_H);
  const std::vector<int> bW = chunk_boundaries((int)(c->nA / 16), c->CH_W);
  c->chunksH = (int)bH.size() - 1;
  c->chunksW = (int)bW.size() - 1;
  for (int** p : {&c->cstartH, &c->cstartW}) {
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
  }
  HIPCHK(hipMalloc(&c->cstartH, sizeof(int) * bH.size()));
  HIPCHK(hipMalloc(&c->cstartW, sizeof(int) * bW.size()));
  HIPCHK(hipMemcpy(c->cstartH, bH.data(), sizeof(int) * bH.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->cstartW, bW.data(), sizeof(int) * bW.size(), hipMemcpyHostToDevice));
  if (getenv("NBMF_DEBUG"))
    fprintf(stderr, "[nbmf] K_pad=%d path=%d: H-pass %d x %d workgroups (chunk %d blocks, %d slots), W-pass %d x %d (chunk %d, %d slots)\n",
            c->KP, c->data_kind, (int)(c->nA / 64), c->chunksH, c->CH_H, slotsH, (int)(c->mA / 64), c->chunksW, c->CH_W, slotsW);
  for (double** p : {&c->slabH, &c->slabW, &c->lossbuf}) {
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
  }
  const size_t fw = (size_t)c->KP * c->mA * sizeof(double), fh = (size_t)c->KP * c->nA * sizeof(double);
  HIPCHK(hipMalloc(&c->slabH, 2 * (size_t)c->chunksH * fh));
  HIPCHK(hipMalloc(&c->slabW, (size_t)c->chunksW * fw));
  HIPCHK(hipMalloc(&c->lossbuf, sizeof(double) * (size_t)c->chunksH * (c->nA / 16 / WG_WAVES)));
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

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int nbmf_abi_version(void) { return 1; }

const char* nbmf_last_error(void) { return g_err.c_str(); }

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
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(NBMF_ERR_HIP, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

  nbmf_ctx* c = new nbmf_ctx();
  c->device = device;
  c->m = m;
  c->n = n;
  c->k = k;
  c->mA = round_up(m, PAD);
  c->nA = round_up(n, PAD);
  c->KB = k <= 16 ? 1 : k <= 32 ? 2 : k <= 64 ? 4 : 8;
  c->KP = 16 * c->KB;
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

  const size_t fw = (size_t)c->KP * c->mA * sizeof(double), fh = (size_t)c->KP * c->nA * sizeof(double);
  HIPCHK(hipMalloc(&c->Wn, fw));
  HIPCHK(hipMalloc(&c->WT, fw));
  HIPCHK(hipMalloc(&c->WG, fw));
  HIPCHK(hipMalloc(&c->Hn, fh));
  HIPCHK(hipMalloc(&c->HT, fh));
  HIPCHK(hipMalloc(&c->HG, fh));
  // (the pass workspaces depend on the storage path: setup_workspaces, called by nbmf_upload)
  HIPCHK(hipMalloc(&c->Pbuf, 2 * fh + 64));
  c->n_prior_blocks = (int)(((size_t)c->KP * c->nA + 255) / 256);
  HIPCHK(hipMalloc(&c->prior, sizeof(double) * 2 * (size_t)c->n_prior_blocks));
  HIPCHK(hipMalloc(&c->scal, sizeof(double) * 8));
  HIPCHK(hipMalloc(&c->sbuf, sizeof(double) * 8));
  HIPCHK(hipMalloc(&c->flags, sizeof(int) * 8));
  HIPCHK(hipMalloc(&c->stats, sizeof(unsigned long long) * 8));
  HIPCHK(hipMalloc(&c->rowcnt, sizeof(double) * (size_t)c->mA));
  HIPCHK(hipMemset(c->flags, 0, sizeof(int) * 8));
  HIPCHK(hipMemset(c->scal, 0, sizeof(double) * 8));
  c->stage_bytes = (fw > fh ? fw : fh);
  HIPCHK(hipMalloc(&c->stage, c->stage_bytes));
  *out = c;
  return NBMF_OK;
}

int nbmf_destroy(nbmf_ctx* c) {
  if (!c) return NBMF_OK;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  if (c->host_buf) hipHostFree(c->host_buf);
  void* ptrs[] = {c->dataA, c->dataB, c->maskA, c->maskB, c->rowcnt, c->Wn, c->WT, c->WG, c->Hn, c->HT, c->HG,
                  c->slabH, c->slabW, c->Pbuf, c->lossbuf, c->prior, c->scal, c->flags, c->losses_d, c->stage, c->stats,
                  c->sbuf, c->Qbuf, c->cstartH, c->cstartW};
  for (void* p : ptrs)
    if (p) hipFree(p);
  for (hipEvent_t e : c->ev) hipEventDestroy(e);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return NBMF_OK;
}

int nbmf_set_hyper(nbmf_ctx* c, double alpha, double beta, double eps, int projection) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (projection != NBMF_PROJ_NORMALIZE && projection != NBMF_PROJ_DUCHI)
    return fail(NBMF_ERR_ARG, "unknown projection %d", projection);
  if (!(eps > 0)) return fail(NBMF_ERR_ARG, "eps must be > 0");
  c->alpha = alpha;
  c->beta = beta;
  c->eps = eps;
  c->projection = projection;
  return NBMF_OK;
}

int nbmf_upload(nbmf_ctx* c, const double* x, int64_t ldx, int transposed, const void* mask, int mask_kind,
                int64_t ldmask, int* out_flags) {
  if (!c || !x) return fail(NBMF_ERR_ARG, "null context or data");
  if (mask_kind != NBMF_MASK_NONE && !mask) return fail(NBMF_ERR_ARG, "mask_kind set but mask is NULL");
  if (!mask) mask_kind = NBMF_MASK_NONE;
  if (int rc = set_device(c)) return rc;
  const int64_t U = transposed ? c->n : c->m, V = transposed ? c->m : c->n;
  if (ldx < V || (mask && ldmask < V)) return fail(NBMF_ERR_ARG, "leading dimension smaller than the row length");

  // cheap host-side guess of the storage path from a sample (the device pack verifies it exactly);
  // NBMF_FORCE_F64=1 keeps binary data on the 8-byte path (measurement only)
  bool guess_bin = !(getenv("NBMF_FORCE_F64") && atoi(getenv("NBMF_FORCE_F64")) != 0);
  if (guess_bin) {
    const int64_t rows = U < 8 ? U : 8;
    for (int64_t u = 0; u < rows && guess_bin; ++u) {
      const int64_t uu = (U - 1) * u / (rows > 1 ? rows - 1 : 1);
      for (int64_t v = 0; v < V && v < 4096; ++v) {
        const double xv = x[uu * ldx + v];
        if (xv != 0.0 && xv != 1.0) {
          guess_bin = false;
          break;
        }
        if (mask_kind == NBMF_MASK_F64) {
          const double mk = ((const double*)mask)[uu * ldmask + v];
          if (mk != 0.0 && mk != 1.0) {
            guess_bin = false;
            break;
          }
        }
      }
    }
  }

  const size_t tiles = (size_t)(c->mA / 16) * (c->nA / 16);
  // staging chunk: <= 256 MiB of raw rows and <= 32768 tile rows (grid.y limit of the pack launch)
  const int64_t chunk_rows_max =
      std::min<int64_t>(32768 * 16, std::max<int64_t>(PAD, ((int64_t)(256ll << 20) / (V * 8)) / PAD * PAD));
  double* raw = nullptr;
  void* rawm = nullptr;
  const size_t msz = mask_kind == NBMF_MASK_F64 ? 8 : 1;
  const int64_t chunk_rows = std::min<int64_t>(round_up(U, PAD), chunk_rows_max);
  HIPCHK(hipMalloc(&raw, (size_t)chunk_rows * V * 8));
  if (mask) HIPCHK(hipMalloc(&rawm, (size_t)chunk_rows * V * msz));

  int rc = NBMF_OK;
  unsigned long long st[4] = {0, 0, 0, 0};
  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool binary = (attempt == 0) ? guess_bin : false;
    for (void** p : {&c->dataA, &c->dataB, &c->maskA, &c->maskB}) {
      if (*p) hipFree(*p);
      *p = nullptr;
    }
    c->data_kind = -1;
    const size_t esz = binary ? 1 : 8;
    const size_t bytes = tiles * 256 * esz;
    HIPCHK(hipMalloc(&c->dataA, bytes));
    HIPCHK(hipMalloc(&c->dataB, bytes));
    // pad tiles must read as "invalid": code 0 on the binary path, -1.0 on the f64 path
    // (the pack kernel writes every tile of the padded mA x nA grid; out-of-range lanes get the
    //  "invalid" marker: code 0 on the binary path, -1.0 on the f64 path)
    if (!binary && mask_kind != NBMF_MASK_NONE) {
      HIPCHK(hipMalloc(&c->maskA, bytes));
      HIPCHK(hipMalloc(&c->maskB, bytes));
    }
    HIPCHK(hipMemsetAsync(c->stats, 0, sizeof(unsigned long long) * 8, c->stream));

    for (int64_t u0 = 0; u0 < round_up(U, PAD); u0 += chunk_rows) {
      const int64_t urows_pad = std::min<int64_t>(chunk_rows, round_up(U, PAD) - u0);
      const int64_t urows = std::max<int64_t>(0, std::min<int64_t>(U - u0, urows_pad));
      if (urows > 0) {
        HIPCHK(hipMemcpy2DAsync(raw, (size_t)V * 8, x + u0 * ldx, (size_t)ldx * 8, (size_t)V * 8, (size_t)urows,
                                hipMemcpyHostToDevice, c->stream));
        if (mask)
          HIPCHK(hipMemcpy2DAsync(rawm, (size_t)V * msz, (const char*)mask + (size_t)u0 * ldmask * msz,
                                  (size_t)ldmask * msz, (size_t)V * msz, (size_t)urows, hipMemcpyHostToDevice,
                                  c->stream));
      }
      PackArgs a{};
      a.x = raw;
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
      a.dataA = c->dataA;
      a.dataB = c->dataB;
      a.maskA = c->maskA;
      a.maskB = c->maskB;
      a.RbA = c->mA / 16;
      a.RbB = c->nA / 16;
      a.stats = c->stats;
      const int64_t VA = transposed ? c->mA : c->nA;
      dim3 grid((unsigned)(VA / 16 / 4), (unsigned)(urows_pad / 16));
      hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, c->stream, a);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(c->stream));   // raw staging buffer is reused by the next chunk
    }
    HIPCHK(hipMemcpy(st, c->stats, sizeof st, hipMemcpyDeviceToHost));
    if (st[1] != 0) {
      rc = fail(NBMF_ERR_RANGE, "X must be binary: %llu entries outside [0,1] or not finite", st[1]);
      break;
    }
    const bool is_bin = (st[2] == 0 && st[3] == 0);
    if (binary && !is_bin) continue;   // sample guessed wrong: repack as doubles
    c->data_kind = binary ? DATA_BIN : (mask_kind != NBMF_MASK_NONE ? DATA_F64M : DATA_F64);
    break;
  }
  hipFree(raw);
  if (rawm) hipFree(rawm);
  if (rc != NBMF_OK) return rc;
  if (c->data_kind < 0) return fail(NBMF_ERR_STATE, "internal: pack did not settle on a storage path");
  if (int rc2 = setup_workspaces(c)) return rc2;
  c->n_obs = (mask_kind == NBMF_MASK_NONE) ? (double)c->m * (double)c->n : (double)st[0];
  c->n_obs_global = c->n_obs;
  hipLaunchKernelGGL(rowcount_kernel, dim3((unsigned)((c->m + 255) / 256)), dim3(256), 0, c->stream, c->dataB, c->maskB,
                     c->data_kind, (long long)(c->nA / 16), (long long)c->m, c->rowcnt);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  if (out_flags) *out_flags = (c->data_kind == DATA_BIN) ? NBMF_FLAG_BINARY_PATH : 0;
  return NBMF_OK;
}

int nbmf_get_n_obs(nbmf_ctx* c, double* n_obs) {
  if (!c || !n_obs) return fail(NBMF_ERR_ARG, "null argument");
  if (c->data_kind < 0) return fail(NBMF_ERR_STATE, "nbmf_upload has not been called");
  *n_obs = c->n_obs;
  return NBMF_OK;
}

int nbmf_set_factors(nbmf_ctx* c, const double* W, const double* H) {
  if (!c || !W || !H) return fail(NBMF_ERR_ARG, "null argument");
  if (int rc = set_device(c)) return rc;
  HIPCHK(hipMemcpyAsync(c->stage, W, sizeof(double) * (size_t)c->k * c->m, hipMemcpyHostToDevice, c->stream));
  {
    const long long tot = (long long)c->KP * c->mA;
    hipLaunchKernelGGL(set_factor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->stage, c->Wn,
                       c->WT, c->WG, c->k, c->KP, (long long)c->m, (long long)c->mA);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpyAsync(c->stage, H, sizeof(double) * (size_t)c->k * c->n, hipMemcpyHostToDevice, c->stream));
  {
    const long long tot = (long long)c->KP * c->nA;
    hipLaunchKernelGGL(set_factor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->stage, c->Hn,
                       c->HT, c->HG, c->k, c->KP, (long long)c->n, (long long)c->nA);
    HIPCHK(hipGetLastError());
  }
  hipLaunchKernelGGL(prior_kernel, dim3(c->n_prior_blocks), dim3(256), 0, c->stream, c->Hn, c->prior, c->k, c->KP,
                     (long long)c->n, (long long)c->nA, c->eps);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
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
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_losses(c, max_iter)) return rc;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->scal, 0, sizeof(double) * 8, c->stream));

  // Timeline (N3 of SURVEY Appendix A): the H-pass of iteration t also yields the log-likelihood of
  // the factors after iteration t-1, so loss(t-1) and its stop test are settled before H-update(t).
  const int batch = (tol > 0.0) ? 8 : max_iter;
  int host_done = 0;
  int it = 0;
  // One iteration = five launches with identical arguments every time (the loss index is counted on the
  // device), so on a single GPU it can be captured once into a hipGraph and replayed.  Measured: no gain
  // (config 1, 100x500 K=6: 14.7k it/s replayed vs 16.5k it/s eager) -- tiny problems are bound by the
  // latency of the five dependent kernels, not by the host's launch rate -- so it is opt-in
  // (NBMF_USE_GRAPH=1), never used with a communicator or event timing.
  const bool use_graph = getenv("NBMF_USE_GRAPH") && !is_sharded(c) && !c->timing && max_iter >= 8;
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
  while (it < max_iter && !host_done) {
    const int end = std::min(max_iter, it + batch);
    for (; it < end; ++it) {
      if (use_graph) {
        HIPCHK(hipGraphLaunch(gexec, c->stream));
        continue;
      }
      if (int rc = enqueue_h_pass(c)) return rc;
      if (it > 0)
        if (int rc = enqueue_finalize(c, it - 1, tol)) return rc;
      if (int rc = enqueue_h_update(c)) return rc;
      if (int rc = enqueue_w_step(c, c->projection)) return rc;
    }
    if (tol > 0.0 && it < max_iter) {
      int fl[2];
      HIPCHK(hipMemcpyAsync(fl, c->flags, sizeof fl, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      if (c->timing) timing_collect(c);
      host_done = fl[0];
    }
  }
  if (!host_done) {
    if (int rc = enqueue_loglik_pass(c, 0)) return rc;            // loss of the last iteration (Theta-only sweep)
    if (int rc = enqueue_finalize(c, use_graph ? -1 : max_iter - 1, tol)) return rc;
  }
  int fl[2];
  HIPCHK(hipMemcpyAsync(fl, c->flags, sizeof fl, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->timing) timing_collect(c);
  const int nit = fl[1];
  if (nit < 1 || nit > max_iter) return fail(NBMF_ERR_STATE, "internal: device reported n_iter=%d", nit);
  HIPCHK(hipMemcpy(losses, c->losses_d, sizeof(double) * (size_t)nit, hipMemcpyDeviceToHost));
  *n_iter = nit;
  return NBMF_OK;
}

int nbmf_w_only_steps(nbmf_ctx* c, int n_steps) {
  if (int rc = ready(c)) return rc;
  if (n_steps < 0) return fail(NBMF_ERR_ARG, "n_steps must be >= 0");
  if (int rc = set_device(c)) return rc;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));
  for (int s = 0; s < n_steps; ++s)
    if (int rc = enqueue_w_step(c, NBMF_PROJ_NORMALIZE)) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->timing) timing_collect(c);
  return NBMF_OK;
}

int nbmf_loss(nbmf_ctx* c, double* loss) {
  if (int rc = ready(c)) return rc;
  if (!loss) return fail(NBMF_ERR_ARG, "null output");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_losses(c, 1)) return rc;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 8, c->stream));
  hipLaunchKernelGGL(prior_kernel, dim3(c->n_prior_blocks), dim3(256), 0, c->stream, c->Hn, c->prior, c->k, c->KP,
                     (long long)c->n, (long long)c->nA, c->eps);
  HIPCHK(hipGetLastError());
  if (int rc = enqueue_loglik_pass(c, 0)) return rc;
  if (int rc = enqueue_finalize(c, 0, 0.0)) return rc;
  HIPCHK(hipMemcpyAsync(loss, c->losses_d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->timing) timing_collect(c);
  return NBMF_OK;
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
  return NBMF_OK;
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
  return NBMF_OK;
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
  c->nranks = nranks;
  c->rank = rank;
  c->shard_axis = shard_axis;
  if (shard_axis == 1 && !c->Qbuf) HIPCHK(hipMalloc(&c->Qbuf, sizeof(double) * (size_t)c->KP * c->mA));
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
    if (int rc = all_reduce_inplace(c, c->rowcnt, (size_t)c->mA)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return NBMF_OK;
}

static int comm_check_args(nbmf_ctx* c, int nranks, int rank, int shard_axis) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(NBMF_ERR_ARG, "bad rank %d / nranks %d", rank, nranks);
  if (shard_axis != 0 && shard_axis != 1) return fail(NBMF_ERR_ARG, "shard_axis must be 0 (rows of Y) or 1 (columns of Y)");
  if (c->data_kind < 0) return fail(NBMF_ERR_STATE, "call nbmf_upload before attaching a communicator (global counts are reduced there)");
  if (c->comm || c->host_reduce) return fail(NBMF_ERR_STATE, "a communicator is already attached");
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

int nbmf_timing_enable(nbmf_ctx* c, int enable) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  c->timing = enable != 0;
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

int nbmf_synchronize(nbmf_ctx* c) {
  if (!c) return fail(NBMF_ERR_ARG, "null context");
  if (int rc = set_device(c)) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return NBMF_OK;
}

int nbmf_selftest_unary(int device, int op, int n, const double* x, double* y) {
  if (!x || !y || n < 1 || (op != 0 && op != 1)) return fail(NBMF_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(device));
  double *d = nullptr, *o = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(double) * (size_t)n));
  HIPCHK(hipMalloc(&o, sizeof(double) * (size_t)n));
  HIPCHK(hipMemcpy(d, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(unary_test_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, d, o, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(y, o, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  hipFree(d);
  hipFree(o);
  return NBMF_OK;
}

}  // extern "C"
