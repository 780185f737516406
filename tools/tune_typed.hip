// tools/tune_typed.hip -- harness: bucketed evaluation, f32 vs f64 at equal row bytes (16 KiB rows, 1e6 queries).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../ndarray-interp_amd/csrc/kernels.hpp"
using namespace ndi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <class F> static double time_ms(F&& launch, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  std::vector<double> ts;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
  std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
template <class T> __global__ void fill_rand(T* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (T)((h >> 8) * (1.0 / 16777216.0));
  }
}
template <class T, int U> static void run(const char* name, uint64_t L, uint64_t Q, void* d_out) {
  const uint64_t n = 4096;
  std::mt19937_64 rng(1);
  std::vector<uint32_t> idx(Q), perm(Q);
  std::vector<T> t(Q);
  for (uint64_t i = 0; i < Q; ++i) { idx[i] = (uint32_t)(rng() % (n - 1)); t[i] = (T)((rng() >> 11) * (1.0 / 9007199254740992.0)); perm[i] = (uint32_t)i; }
  std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return idx[a] < idx[b]; });
  T *data, *ca, *cb, *dt, *knots; uint32_t *didx; uint4* drec; StatusBlock* st;
  CK(hipMalloc(&data, n * L * sizeof(T))); CK(hipMalloc(&ca, n * L * sizeof(T))); CK(hipMalloc(&cb, n * L * sizeof(T)));
  CK(hipMalloc(&dt, Q * sizeof(T))); CK(hipMalloc(&didx, Q * 4)); CK(hipMalloc(&drec, Q * 16)); CK(hipMalloc(&st, sizeof(StatusBlock)));
  CK(hipMalloc(&knots, n * sizeof(T)));
  hipLaunchKernelGGL(fill_rand<T>, dim3(4096), dim3(256), 0, 0, data, n * L, 1u);
  hipLaunchKernelGGL(fill_rand<T>, dim3(4096), dim3(256), 0, 0, ca, n * L, 2u);
  hipLaunchKernelGGL(fill_rand<T>, dim3(4096), dim3(256), 0, 0, cb, n * L, 3u);
  CK(hipMemcpy(dt, t.data(), Q * sizeof(T), hipMemcpyHostToDevice)); CK(hipMemcpy(didx, idx.data(), Q * 4, hipMemcpyHostToDevice));
  {   // grouped records {query index, interval, t bits} as the library's scatter pass writes them
    std::vector<uint4> rec(Q);
    for (uint64_t p = 0; p < Q; ++p) {
      const uint32_t qi = perm[p];
      if (sizeof(T) == 8) { unsigned long long b; memcpy(&b, &t[qi], 8); rec[p] = make_uint4(qi, idx[qi], (uint32_t)b, (uint32_t)(b >> 32)); }
      else { uint32_t b; memcpy(&b, &t[qi], 4); rec[p] = make_uint4(qi, idx[qi], b, 0u); }
    }
    CK(hipMemcpy(drec, rec.data(), Q * 16, hipMemcpyHostToDevice));
  }
  CK(hipMemset(st, 0xFF, 16)); CK(hipMemset((char*)st + 16, 0, sizeof(StatusBlock) - 16));
  Eval1Args<T> A{}; A.knots = knots; A.data = data; A.ca = ca; A.cb = cb; A.q = dt; A.idx = didx; A.t = dt; A.out = (T*)d_out;
  A.lanes = L; A.out_stride = L; A.nq = Q; A.status = st; A.rec = drec; A.run = 1;
  const uint64_t LV = L / Wide<T>::N;
  const unsigned segs = (unsigned)((LV + 256 * U - 1) / (256 * U));
  const uint64_t per = ((Q + 127) / 128 + 7) / 8;
  const double out_gb = Q * L * sizeof(T) / 1e9;
  for (int rep = 0; rep < 2; ++rep) {
    double t1 = time_ms([&] { hipLaunchKernelGGL((eval_bucketed_kernel<T, ST_CUBIC, U, 128, true>), dim3((unsigned)(per * 8), segs), dim3(256), 0, 0, A); });
    double t2 = time_ms([&] { hipLaunchKernelGGL((eval_bucketed_kernel<T, ST_LINEAR, U, 128, true>), dim3((unsigned)(per * 8), segs), dim3(256), 0, 0, A); });
    printf("%s U=%d L=%llu  cubic %7.3f ms (%5.0f GB/s out)   linear-formula %7.3f ms (%5.0f GB/s out)\n", name, U, (unsigned long long)L, t1, out_gb / t1 * 1e3, t2, out_gb / t2 * 1e3);
  }
  CK(hipFree(data)); CK(hipFree(ca)); CK(hipFree(cb)); CK(hipFree(dt)); CK(hipFree(didx)); CK(hipFree(drec)); CK(hipFree(st)); CK(hipFree(knots));
}
int main() {
  const uint64_t Q = 1000000;
  void* out; CK(hipMalloc(&out, Q * 4096 * 4));   // 16.4 GB, shared by all runs (same placement)
  run<double, 4>("f64", 2048, Q, out);
  run<float, 4>("f32", 4096, Q, out);
  run<double, 2>("f64", 2048, Q, out);
  run<float, 2>("f32", 4096, Q, out);
  run<double, 1>("f64", 2048, Q, out);
  run<float, 1>("f32", 4096, Q, out);
  return 0;
}
