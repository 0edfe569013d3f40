// Accuracy of the hardware reciprocal seed v_rcp_f64 on gfx950, and of one quadratic / one cubic Newton step on it.
// build: hipcc -O3 --offload-arch=gfx950 -o build/microbench_rcp tools/microbench_rcp.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* seed, double* quad, double* cub, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  const double r = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, r, 1.0);
  seed[i] = r;
  quad[i] = __builtin_fma(r, e, r);
  cub[i] = __builtin_fma(r, __builtin_fma(e, e, e), r);
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (s >> 11) * (1.0 / 9007199254740992.0);
    // half of the samples log-uniform in [1e-8, 1], half uniform in [0, 1] (+eps as the kernel sees them)
    x[i] = (i & 1) ? std::exp(std::log(1e-8) * u) : u + 1e-8;
  }
  double *dx, *da, *db, *dc;
  hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, dc, n);
  hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
  double ms = 0, mq = 0, mc = 0;
  long long nq = 0, nc = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)x[i];
    const double ulp = std::ldexp(1.0, std::ilogb((double)t) - 52);
    ms = std::fmax(ms, (double)(fabsl((long double)a[i] - t) / t));
    const double eq = (double)(fabsl((long double)b[i] - t) / ulp), ec = (double)(fabsl((long double)c[i] - t) / ulp);
    mq = std::fmax(mq, eq); mc = std::fmax(mc, ec);
    nq += b[i] != (double)t; nc += c[i] != (double)t;
  }
  printf("seed: max relative error %.3g (2^%.1f)\n", ms, std::log2(ms));
  printf("quadratic step (2 FMAs): max %.3f ulp, %lld of %d differ from the correctly rounded 1/x\n", mq, nq, n);
  printf("cubic step (3 FMAs):     max %.3f ulp, %lld of %d differ\n", mc, nc, n);
  return 0;
}
