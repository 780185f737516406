// tools/placement_probe.hip -- why does the same write stream run 5.0 ... 6.5 ms into different 32.8 GB
// allocations?  (VERDICT r1 weak #3 / next #5.)  Emulates the bucketed evaluation's output stream -- every
// workgroup writes whole 32 KiB rows at scattered row positions with non-temporal 16-byte stores -- and
// measures, per allocation:
//   seq        sequential fill of the whole buffer
//   pieces     sequential fill of every 1 GiB piece on its own (is the slowness localized?)
//   scatter    random row order, row stride 32 KiB
//   pad128/pad256/pad4k   same with a padded row stride (breaks the power-of-two stride)
//   rot        same, the starting 4 KiB segment of a row rotates per workgroup
//   win1g      random order inside 1 GiB windows that are visited one after the other
// for plain hipMalloc buffers, a hipDeviceMallocContiguous buffer, a hipMallocAsync buffer and a buffer mapped
// from 1 GiB physical chunks (hipMemCreate / hipMemMap).  Prints one JSON object per line.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e = (x);                                                                    \
    if (e != hipSuccess) {                                                                 \
      printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e), __LINE__);          \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

constexpr uint32_t NROWS = 1000000;
constexpr uint32_t ROW_VECS = 2048;                 // 4096 f64 = 32 KiB
constexpr uint32_t CQ = 128;                        // rows per workgroup, as eval_bucketed_kernel

__global__ __launch_bounds__(256) void fill_kernel(dbl2* p, size_t nvec, double v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    dbl2 x = {v, v + 1.0};
    __builtin_nontemporal_store(x, p + i);
  }
}

// order[]: row visited at grouped position p; stride_vecs: row pitch in 16-byte vectors; rot: rotate the
// starting segment by the workgroup's id
__global__ __launch_bounds__(256) void scatter_kernel(dbl2* out, const uint32_t* order, uint32_t nrows,
                                                      uint64_t stride_vecs, int rot, double v) {
  __shared__ uint32_t s_row[CQ];
  const uint32_t nchunks = (nrows + CQ - 1) / CQ;
  const uint32_t per = (nchunks + 7) / 8;
  for (uint32_t vb = blockIdx.x; vb < per * 8; vb += gridDim.x) {
    const uint32_t chunk = (vb & 7u) * per + (vb >> 3);   // same XCD-aware chunk order as the library kernel
    if (chunk >= nchunks) continue;
    const uint32_t p0 = chunk * CQ;
    const uint32_t cnt = min(CQ, nrows - p0);
    __syncthreads();
    if (threadIdx.x < cnt) s_row[threadIdx.x] = order[p0 + threadIdx.x];
    __syncthreads();
    const uint32_t u0 = rot ? (chunk & 7u) : 0u;
    for (uint32_t j = 0; j < cnt; ++j) {
      dbl2* o = out + (uint64_t)s_row[j] * stride_vecs;
      const dbl2 x = {v + j, v};
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) __builtin_nontemporal_store(x, o + (((u + u0) & 7u) * 256u + threadIdx.x));
    }
  }
}

static float median_ms(std::vector<float> t) {
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

template <class F>
static float time_it(F&& launch, int reps = 3) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();  // warm
  std::vector<float> ts;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ts.push_back(ms);
  }
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return median_ms(ts);
}

struct Orders {
  uint32_t* random;
  uint32_t* win1g;
};

static void probe(const char* kind, int id, void* p, size_t bytes, const Orders& od, bool pieces) {
  const size_t out_bytes = (size_t)NROWS * ROW_VECS * 16;
  const float seq = time_it([&] { hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, (dbl2*)p, out_bytes / 16, 1.0); });
  auto scat = [&](const uint32_t* order, uint64_t stride, int rot) {
    return time_it([&] {
      hipLaunchKernelGGL(scatter_kernel, dim3(7816), dim3(256), 0, 0, (dbl2*)p, order, NROWS, stride, rot, 2.0);
    });
  };
  const float s0 = scat(od.random, ROW_VECS, 0);
  const float p128 = scat(od.random, ROW_VECS + 8, 0);
  const float p256 = scat(od.random, ROW_VECS + 16, 0);
  const float p4k = scat(od.random, ROW_VECS + 256, 0);
  const float rot = scat(od.random, ROW_VECS, 1);
  const float win = scat(od.win1g, ROW_VECS, 0);
  printf("{\"alloc\": \"%s\", \"id\": %d, \"ptr\": \"%p\", \"seq_ms\": %.3f, \"seq_TBs\": %.2f, \"scatter_ms\": %.3f, "
         "\"scatter_TBs\": %.2f, \"pad128_ms\": %.3f, \"pad256_ms\": %.3f, \"pad4k_ms\": %.3f, \"rot_ms\": %.3f, "
         "\"win1g_ms\": %.3f}\n",
         kind, id, p, seq, out_bytes / seq / 1e9, s0, out_bytes / s0 / 1e9, p128, p256, p4k, rot, win);
  if (pieces) {
    const size_t piece = 1ull << 30;
    std::string s = "{\"alloc\": \"" + std::string(kind) + "\", \"id\": " + std::to_string(id) + ", \"piece_GBs\": [";
    for (size_t off = 0; off + piece <= out_bytes; off += piece) {
      const float t = time_it([&] {
        hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, (dbl2*)((char*)p + off), piece / 16, 3.0);
      }, 2);
      char buf[32];
      snprintf(buf, sizeof buf, "%s%.0f", off ? ", " : "", piece / t / 1e6);
      s += buf;
    }
    printf("%s]}\n", s.c_str());
  }
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 5;
  const size_t bytes = (size_t)NROWS * (ROW_VECS + 256) * 16;   // room for the largest padded stride (36.9 GB)
  std::mt19937 rng(7);
  std::vector<uint32_t> h(NROWS);
  for (uint32_t i = 0; i < NROWS; ++i) h[i] = i;
  std::shuffle(h.begin(), h.end(), rng);
  Orders od{};
  CK(hipMalloc(&od.random, NROWS * 4));
  CK(hipMemcpy(od.random, h.data(), NROWS * 4, hipMemcpyHostToDevice));
  for (uint32_t i = 0; i < NROWS; ++i) h[i] = i;
  const uint32_t W = 32768;   // rows per 1 GiB window
  for (uint32_t w0 = 0; w0 < NROWS; w0 += W) std::shuffle(h.begin() + w0, h.begin() + std::min(NROWS, w0 + W), rng);
  CK(hipMalloc(&od.win1g, NROWS * 4));
  CK(hipMemcpy(od.win1g, h.data(), NROWS * 4, hipMemcpyHostToDevice));

  std::vector<void*> bufs(K, nullptr);
  for (int i = 0; i < K; ++i) CK(hipMalloc(&bufs[i], bytes));
  for (int i = 0; i < K; ++i) probe("hipMalloc", i, bufs[i], bytes, od, true);
  for (int i = 0; i < K; ++i) probe("hipMalloc-again", i, bufs[i], bytes, od, false);   // stable per buffer?
  for (int i = 1; i < K; ++i) CK(hipFree(bufs[i]));   // keep buffer 0 so the next allocations land elsewhere

  {
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous);
    if (e == hipSuccess) {
      probe("contiguous", 0, p, bytes, od, true);
      CK(hipFree(p));
    } else {
      printf("{\"alloc\": \"contiguous\", \"error\": \"%s\"}\n", hipGetErrorString(e));
      (void)hipGetLastError();
    }
  }
  {
    void* p = nullptr;
    hipError_t e = hipMallocAsync(&p, bytes, 0);
    if (e == hipSuccess) {
      CK(hipStreamSynchronize(0));
      probe("hipMallocAsync", 0, p, bytes, od, false);
      CK(hipFreeAsync(p, 0));
      CK(hipStreamSynchronize(0));
    } else {
      printf("{\"alloc\": \"hipMallocAsync\", \"error\": \"%s\"}\n", hipGetErrorString(e));
      (void)hipGetLastError();
    }
  }
  {  // virtual-memory API: the buffer is mapped from separately created 1 GiB physical chunks
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e == hipSuccess) {
      const size_t chunk = 1ull << 30;
      const size_t total = (bytes + chunk - 1) / chunk * chunk;
      void* va = nullptr;
      CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
      std::vector<hipMemGenericAllocationHandle_t> hs;
      for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t hnd;
        CK(hipMemCreate(&hnd, chunk, &prop, 0));
        CK(hipMemMap((char*)va + off, chunk, 0, hnd, 0));
        hs.push_back(hnd);
      }
      hipMemAccessDesc acc{};
      acc.location = prop.location;
      acc.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(va, total, &acc, 1));
      printf("{\"alloc\": \"vmm-1GiB-chunks\", \"granularity\": %zu}\n", gran);
      probe("vmm-1GiB-chunks", 0, va, bytes, od, true);
      CK(hipMemUnmap(va, total));
      for (auto hnd : hs) CK(hipMemRelease(hnd));
      CK(hipMemAddressFree(va, total));
    } else {
      printf("{\"alloc\": \"vmm\", \"error\": \"%s\"}\n", hipGetErrorString(e));
      (void)hipGetLastError();
    }
  }
  for (int i = 1; i < K; ++i) {   // fresh allocations after the frees
    void* p = nullptr;
    CK(hipMalloc(&p, bytes));
    probe("hipMalloc-realloc", i, p, bytes, od, false);
    CK(hipFree(p));
  }
  CK(hipFree(bufs[0]));
  return 0;
}
