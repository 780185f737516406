// tools/output_alloc_probe.hip -- round 6: can a library-owned output allocation be made placement-robust WITHOUT holding most
// of the device memory?  A 32.8 GB buffer (BASELINE configs[1]'s output) assembled from CHUNK-GiB physical chunks
// (hipMemCreate) of which every K-th of K * need sequentially created chunks is kept (the others are released again): the kept
// chunks span K x the buffer's size of the physical range.  Measures creation time and the scattered-row store stream of
// the bucketed evaluation (placement_probe2's scatter_kernel) into each, next to plain hipMalloc buffers.
//   ./output_alloc_probe [chunk_GiB=1] [reps_per_variant=2]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr uint32_t NROWS = 1000000, ROW_VECS = 2048, CQ = 128;

__global__ __launch_bounds__(256) void scatter_kernel(dbl2* out, const uint32_t* order, uint32_t nrows, double v) {
  __shared__ uint32_t s_row[CQ];
  const uint32_t nchunks = (nrows + CQ - 1) / CQ, per = (nchunks + 7) / 8;
  for (uint32_t vb = blockIdx.x; vb < per * 8; vb += gridDim.x) {
    const uint32_t chunk = (vb & 7u) * per + (vb >> 3);
    if (chunk >= nchunks) continue;
    const uint32_t p0 = chunk * CQ, cnt = min(CQ, nrows - p0);
    __syncthreads();
    if (threadIdx.x < cnt) s_row[threadIdx.x] = order[p0 + threadIdx.x];
    __syncthreads();
    for (uint32_t j = 0; j < cnt; ++j) {
      dbl2* o = out + (uint64_t)s_row[j] * ROW_VECS;
      const dbl2 x = {v + j, v};
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) __builtin_nontemporal_store(x, o + (u * 256u + threadIdx.x));
    }
  }
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static float scatter_ms(void* p, const uint32_t* order) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(scatter_kernel, dim3(7816), dim3(256), 0, 0, (dbl2*)p, order, NROWS, 2.0);
  // WARM_MS: sustained load before the timed launches (is the spread a clock / power-state ramp rather than placement?)
  if (const char* w = getenv("WARM_MS")) {
    const double t_end = now_s() + atof(w) * 1e-3;
    while (now_s() < t_end) {
      hipLaunchKernelGGL(scatter_kernel, dim3(7816), dim3(256), 0, 0, (dbl2*)p, order, NROWS, 2.0);
      CK(hipDeviceSynchronize());
    }
  }
  std::vector<float> ts;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(scatter_kernel, dim3(7816), dim3(256), 0, 0, (dbl2*)p, order, NROWS, 2.0);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ts.push_back(ms);
  }
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  std::sort(ts.begin(), ts.end());
  return ts[1];
}


int main(int argc, char** argv) {
  const size_t chunk_gib = argc > 1 ? atoi(argv[1]) : 1;
  const int reps = argc > 2 ? atoi(argv[2]) : 2;
  const size_t chunk = chunk_gib << 30;
  const size_t out_bytes = (size_t)NROWS * ROW_VECS * 16;
  const int need = (int)((out_bytes + chunk - 1) / chunk);
  std::mt19937 rng(7);
  std::vector<uint32_t> h(NROWS);
  std::iota(h.begin(), h.end(), 0u);
  std::shuffle(h.begin(), h.end(), rng);
  uint32_t* order;
  CK(hipMalloc(&order, NROWS * 4));
  CK(hipMemcpy(order, h.data(), NROWS * 4, hipMemcpyHostToDevice));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("{\"granularity\": %zu, \"chunk_GiB\": %zu, \"need_chunks\": %d}\n", gran, chunk_gib, need);
  for (int rep = 0; rep < reps; ++rep) {
    std::vector<int> Ks = {0, 1, 2, 3, 4, 6};
    if (const char* ke = getenv("KS")) { Ks.clear(); for (const char* c = ke; *c; ++c) if (*c >= '0' && *c <= '9') Ks.push_back(*c - '0'); }
    const size_t va_align = getenv("ALIGN") ? (size_t)atoll(getenv("ALIGN")) : 0;
    for (int K : Ks) {
      size_t fr = 0, tot = 0;
      CK(hipMemGetInfo(&fr, &tot));
      if (K == 0) {
        void* p;
        const double t0 = now_s();
        CK(hipMalloc(&p, out_bytes));
        const double t1 = now_s();
        const float ms = scatter_ms(p, order);
        printf("{\"rep\": %d, \"alloc\": \"hipMalloc\", \"alloc_ms\": %.1f, \"scatter_ms\": %.3f, \"TBs\": %.2f, \"free_GB\": %.0f, \"va\": \"%p\"}\n", rep,
               (t1 - t0) * 1e3, ms, out_bytes / ms / 1e9, fr / 1e9, p);
        fflush(stdout);
        CK(hipFree(p));
        continue;
      }
      const int total = need * K;
      if ((size_t)total * chunk + (8ull << 30) > fr) {
        printf("{\"rep\": %d, \"K\": %d, \"skipped\": \"needs %zu GB, %zu free\"}\n", rep, K, (size_t)total * chunk >> 30, fr >> 30);
        continue;
      }
      const double t0 = now_s();
      std::vector<hipMemGenericAllocationHandle_t> hs(total);
      for (int i = 0; i < total; ++i) CK(hipMemCreate(&hs[i], chunk, &prop, 0));
      const double t1 = now_s();
      for (int i = 0; i < total; ++i)
        if (i % K != K / 2) CK(hipMemRelease(hs[i]));       // keep one chunk of every K
      void* va = nullptr;
      CK(hipMemAddressReserve(&va, (size_t)need * chunk, va_align, nullptr, 0));
      for (int k = 0; k < need; ++k) CK(hipMemMap((char*)va + (size_t)k * chunk, chunk, 0, hs[k * K + K / 2], 0));
      CK(hipMemSetAccess(va, (size_t)need * chunk, &acc, 1));
      const double t2 = now_s();
      const float ms = scatter_ms(va, order);
      printf("{\"rep\": %d, \"alloc\": \"vmm\", \"K\": %d, \"create_ms\": %.1f, \"release_map_ms\": %.1f, \"scatter_ms\": %.3f, \"TBs\": %.2f, "
             "\"free_GB\": %.0f, \"va\": \"%p\", \"va_align\": %zu}\n", rep, K, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ms, out_bytes / ms / 1e9, fr / 1e9, va, va_align);
      fflush(stdout);
      CK(hipMemUnmap(va, (size_t)need * chunk));
      CK(hipMemAddressFree(va, (size_t)need * chunk));
      for (int k = 0; k < need; ++k) CK(hipMemRelease(hs[k * K + K / 2]));
    }
  }
  return 0;
}
