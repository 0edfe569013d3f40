// Read / write bandwidth of one GPU's own HBM through the three kinds of device allocation the peer
// transport could use for its arena: ordinary (coarse-grained, L2-cached), fine-grained, uncached.
// build: hipcc -O3 --offload-arch=gfx950 -o build/microbench_uc tools/microbench_uc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void rd8(const double* __restrict__ p, double* out, long long n) {
  double s = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678) *out = s;
}
__global__ void rd16(const double2* __restrict__ p, double* out, long long n2) {
  double s = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
    double2 v = p[i];
    s += v.x + v.y;
  }
  if (s == 12345.678) *out = s;
}
__global__ void rd16nt(const double2* __restrict__ p, double* out, long long n2) {
  double s = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
    double x = __builtin_nontemporal_load(&p[i].x), y = __builtin_nontemporal_load(&p[i].y);
    s += x + y;
  }
  if (s == 12345.678) *out = s;
}
__global__ void wr8(double* __restrict__ p, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = (double)i;
}

int main() {
  const long long n = 1 << 20;   // 8 MiB
  double* out;
  CHK(hipMalloc(&out, 8));
  const char* names[] = {"ordinary", "fine-grained", "uncached"};
  for (int kind = 0; kind < 3; ++kind) {
    double* p = nullptr;
    if (kind == 0) CHK(hipMalloc(&p, n * 8));
    if (kind == 1) CHK(hipExtMallocWithFlags((void**)&p, n * 8, hipDeviceMallocFinegrained));
    if (kind == 2) CHK(hipExtMallocWithFlags((void**)&p, n * 8, hipDeviceMallocUncached));
    CHK(hipMemset(p, 0, n * 8));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int grid : {256, 1024, 4096}) {
      for (int which = 0; which < 4; ++which) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(a);
          if (which == 0) hipLaunchKernelGGL(rd8, dim3(grid), dim3(256), 0, 0, p, out, n);
          if (which == 1) hipLaunchKernelGGL(rd16, dim3(grid), dim3(256), 0, 0, (const double2*)p, out, n / 2);
          if (which == 2) hipLaunchKernelGGL(rd16nt, dim3(grid), dim3(256), 0, 0, (const double2*)p, out, n / 2);
          if (which == 3) hipLaunchKernelGGL(wr8, dim3(grid), dim3(256), 0, 0, p, n);
          hipEventRecord(b);
          hipEventSynchronize(b);
          float ms;
          hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
        }
        const char* w[] = {"read 8B/lane", "read 16B/lane", "read 16B nontemporal", "write 8B/lane"};
        printf("%-13s grid %5d %-22s %8.1f us  %7.1f GB/s\n", names[kind], grid, w[which], best * 1e3, n * 8 / (best * 1e-3) / 1e9);
      }
    }
    hipFree(p);
  }
  return 0;
}
