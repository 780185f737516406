// tools/alloc_probe.hip -- is the streaming-store rate tied to the buffer (physical placement) or to the process?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void fill_kernel(dbl2* p, size_t nvec, double v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    dbl2 x = {v, v + 1.0};
    __builtin_nontemporal_store(x, p + i);
  }
}
static double time_fill(void* p, size_t bytes) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ts;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, (dbl2*)p, bytes / 16, 1.0);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[1];
}
int main() {
  const size_t bytes = 1000000ull * 4096 * 8;
  void* p[6];
  for (int i = 0; i < 6; ++i) CK(hipMalloc(&p[i], bytes));
  for (int round = 0; round < 2; ++round)
    for (int i = 0; i < 6; ++i) { double t = time_fill(p[i], bytes); printf("round %d buffer %d @%p  %.3f ms  %.0f GB/s\n", round, i, p[i], t, bytes / t / 1e6); }
  // sub-ranges of one buffer: first / second half
  for (int h = 0; h < 2; ++h) { double t = time_fill((char*)p[0] + h * (bytes / 2), bytes / 2); printf("buffer 0 half %d  %.3f ms  %.0f GB/s\n", h, t, bytes / 2 / t / 1e6); }
  for (int i = 0; i < 6; ++i) CK(hipFree(p[i]));
  for (int i = 0; i < 3; ++i) { CK(hipMalloc(&p[i], bytes)); double t = time_fill(p[i], bytes); printf("realloc %d @%p  %.3f ms  %.0f GB/s\n", i, p[i], t, bytes / t / 1e6); }
  return 0;
}
