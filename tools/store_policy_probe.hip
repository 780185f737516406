// tools/store_policy_probe.hip -- does the cache policy of the output stores matter?  The evaluation kernels write
// every output row once (32 KiB rows at scattered positions of a large buffer) with non-temporal 16-byte stores.
// gfx950 stores carry three policy bits (sc0, sc1 = coherence scope, nt = non-temporal); this probe streams the same
// scattered-row pattern into one buffer with every combination and prints the rate.  One JSON object per line.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int POLICY>
__device__ __forceinline__ void store16(dbl2* p, dbl2 v) {
  if (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// one workgroup per row visit, rows in a pseudo-random order (multiplicative hash of the row counter), 2048 vectors
// of 16 B per row = 32 KiB: the output pattern of eval_bucketed_kernel at 4096 f64 lanes
template <int POLICY>
__global__ __launch_bounds__(256) void rows_kernel(dbl2* out, uint64_t nrows, uint64_t mult) {
  for (uint64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
    const uint64_t row = (r * mult) % nrows;
    dbl2* o = out + row * 2048;
    const dbl2 v = {(double)r, (double)threadIdx.x};
#pragma unroll
    for (int u = 0; u < 8; ++u) store16<POLICY>(o + u * 256 + threadIdx.x, v);
  }
}

template <int POLICY>
static float run(dbl2* buf, uint64_t nrows, uint64_t mult) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<float> ts;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(rows_kernel<POLICY>, dim3(65528), dim3(256), 0, 0, buf, nrows, mult);
    CK(hipGetLastError());
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[1];
}

int main(int argc, char** argv) {
  const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 76;   // buffer size in GiB
  const uint64_t nrows = gib * (1ull << 30) / 32768;
  const uint64_t mult = 2654435761ull | 1ull;   // odd: a permutation of the rows when nrows is a power of two; close enough otherwise
  dbl2* buf;
  CK(hipMalloc(&buf, nrows * 32768));
  const char* names[8] = {"none", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  for (int pass = 0; pass < 2; ++pass) {
    float ms[8];
    ms[0] = run<0>(buf, nrows, mult); ms[1] = run<1>(buf, nrows, mult); ms[2] = run<2>(buf, nrows, mult);
    ms[3] = run<3>(buf, nrows, mult); ms[4] = run<4>(buf, nrows, mult); ms[5] = run<5>(buf, nrows, mult);
    ms[6] = run<6>(buf, nrows, mult); ms[7] = run<7>(buf, nrows, mult);
    for (int p = 0; p < 8; ++p)
      printf("{\"pass\": %d, \"policy\": \"%s\", \"GiB\": %llu, \"ms\": %.3f, \"TB_s\": %.3f}\n", pass, names[p],
             (unsigned long long)gib, ms[p], nrows * 32768.0 / ms[p] / 1e9);
    fflush(stdout);
  }
  return 0;
}
