// Litmus test for the cross-workgroup hand-off of the single-launch path (nbmf_small_kernel.inc):
// is "every handed-off word stored sc1 and loaded sc1, each storing wave drains its stores, the workgroup meets, ONE lane
// publishes an epoch word, consumers poll the epoch words with sc1 loads" -- NO cache-maintenance fence anywhere -- enough
// for a consumer on another CU / another XCD to see the producer's data on this chip and driver?
//
// G workgroups (one per CU, spread over the 8 XCDs by the dispatcher), T rounds.  Round t: workgroup g stores N doubles
// f(g, t) into its slot, drains, meets, publishes epoch 2t+1; everybody polls all epochs, then reads the slots of R other
// workgroups and compares with f(j, t) -- a stale read returns f(j, t-1) or older -- and counts mismatches; a second
// barrier (epoch 2t+2) keeps round t+1's stores behind round t's loads.  Variants of how data is stored / loaded:
//   0  plain stores, plain loads                      (control: expected to SHOW stale reads, or the test sees nothing)
//   1  sc1 stores,  plain loads
//   2  plain stores, sc1 loads
//   3  sc1 stores,  sc1 loads                          <- the protocol of nbmf_small_kernel.inc: must be 0
//   4  plain stores + agent-scope release before the epoch, agent-scope acquire behind the poll + plain loads (textbook)
// Every poll is bounded (2 s): a lost workgroup ends the kernel instead of hanging the GPU.
// build: hipcc -O3 --offload-arch=gfx950 tools/litmus_handoff.hip -o build/litmus_handoff ; run: build/litmus_handoff [G] [T] [N]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                        \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);     \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ double f_of(int g, int t, int i) { return (double)g * 1048576.0 + (double)t * 1.0 + (double)i * 1e-3; }

__device__ __forceinline__ bool wait_all(unsigned long long* sync, int G, unsigned long long epoch, unsigned long long* abort_word) {
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if (threadIdx.x < 64) {
    const unsigned long long t0 = wall_clock64();
    unsigned polls = 0;
    for (int j = threadIdx.x; j < G; j += 64) {
      while (__hip_atomic_load(sync + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        if ((++polls & 63u) == 0u && (wall_clock64() - t0 > 200000000ull || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(abort_word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
  }
  __syncthreads();
  return ok != 0;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void litmus(double* buf, unsigned long long* sync, unsigned long long* result, int G, int T, int N,
                                                 int R) {
  extern __shared__ double pad[];   // (large dynamic LDS: one workgroup per CU)
  const int g = blockIdx.x;
  unsigned long long stale = 0, reads = 0;
  unsigned long long* abort_word = sync + G;
  if (threadIdx.x == 0) result[4 + g] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
  for (int t = 1; t <= T; ++t) {
    double* mine = buf + (size_t)g * N;
    for (int i = threadIdx.x; i < N; i += 256) {
      if (MODE == 1 || MODE == 3)
        __hip_atomic_store(mine + i, f_of(g, t, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store ... sc1
      else
        mine[i] = f_of(g, t, i);
    }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(sync + g, 2ull * t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!wait_all(sync, G, 2ull * t + 1, abort_word)) break;
    if (MODE == 4) {
      if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
    }
    for (int r = 1; r <= R; ++r) {
      const int j = (g + r * 37) % G;
      const double* theirs = buf + (size_t)j * N;
      for (int i = threadIdx.x; i < N; i += 256) {
        const double v = (MODE == 2 || MODE == 3) ? __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)   // global_load ... sc1
                                                  : *(const volatile double*)(theirs + i);
        stale += v != f_of(j, t, i);
        ++reads;
      }
    }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(sync + g, 2ull * t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!wait_all(sync, G, 2ull * t + 2, abort_word)) break;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    stale += __shfl_xor(stale, off, 64);
    reads += __shfl_xor(reads, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&result[0], stale);
    atomicAdd(&result[1], reads);
  }
}

template <int MODE>
void run(const char* what, int G, int T, int N, int R) {
  double* buf;
  unsigned long long *sync, *result;
  CHK(hipMalloc(&buf, sizeof(double) * (size_t)G * N));
  CHK(hipMalloc(&sync, sizeof(unsigned long long) * (G + 8)));
  CHK(hipMalloc(&result, sizeof(unsigned long long) * (G + 8)));
  CHK(hipMemset(buf, 0, sizeof(double) * (size_t)G * N));
  CHK(hipMemset(sync, 0, sizeof(unsigned long long) * (G + 8)));
  CHK(hipMemset(result, 0, sizeof(unsigned long long) * (G + 8)));
  const size_t lds = 96 * 1024;
  CHK(hipFuncSetAttribute((const void*)litmus<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(litmus<MODE>, dim3(G), dim3(256), lds, 0, buf, sync, result, G, T, N, R);
  CHK(hipGetLastError());
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(G + 8), s(G + 8);
  CHK(hipMemcpy(h.data(), result, sizeof(unsigned long long) * (G + 8), hipMemcpyDeviceToHost));
  CHK(hipMemcpy(s.data(), sync, sizeof(unsigned long long) * (G + 8), hipMemcpyDeviceToHost));
  int xcd[8] = {0};
  for (int g = 0; g < G; ++g) xcd[h[4 + g] & 7]++;
  printf("mode %d  %-52s stale %12llu of %12llu reads (%.3g)  %7.2f us/round%s   workgroups per XCD:", MODE, what, h[0], h[1],
         h[1] ? (double)h[0] / (double)h[1] : 0.0, 1e3 * ms / T, s[G] ? "  ABORTED (a poll timed out)" : "");
  for (int i = 0; i < 8; ++i) printf(" %d", xcd[i]);
  printf("\n");
  CHK(hipFree(buf));
  CHK(hipFree(sync));
  CHK(hipFree(result));
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 128, T = argc > 2 ? atoi(argv[2]) : 20000, N = argc > 3 ? atoi(argv[3]) : 512, R = 4;
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  printf("%s, %d CUs; %d workgroups x %d rounds x %d doubles handed off, %d slots read per workgroup and round\n", prop.gcnArchName,
         prop.multiProcessorCount, G, T, N, R);
  run<0>("plain stores, plain loads (control)", G, T, N, R);
  run<1>("sc1 stores, plain loads", G, T, N, R);
  run<2>("plain stores, sc1 loads", G, T, N, R);
  run<3>("sc1 stores, sc1 loads (the single-launch path)", G, T, N, R);
  run<4>("plain + agent release / acquire fences (textbook)", G, T, N, R);
  return 0;
}
