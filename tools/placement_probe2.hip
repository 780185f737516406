// tools/placement_probe2.hip -- follow-up to placement_probe: is the write rate a property of the individual
// physical chunk?  Creates N physical chunks (hipMemCreate, CHUNK_GiB each), maps each at its own virtual
// address, measures every chunk's sustained streaming-store rate (the chunk is filled PASSES times inside one
// launch, so launch ramps do not matter), then
//   * prints the per-chunk rates (is the distribution bimodal?),
//   * maps the fastest / the slowest 32.8 GB worth of chunks into one contiguous virtual range each and runs the
//     scattered-row stream of the bucketed evaluation over both (does chunk selection carry over?),
//   * repeats the per-chunk measurement on the 4 GiB pieces of plain hipMalloc buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <string>
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

__global__ __launch_bounds__(256) void fill_passes(dbl2* p, size_t nvec, int passes, double v) {
  for (int k = 0; k < passes; ++k)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
      dbl2 x = {v + k, v};
      __builtin_nontemporal_store(x, p + i);
    }
}

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

template <class F>
static float time_it(F&& launch, int reps = 3) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
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
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

static double chunk_rate(void* p, size_t bytes, int passes) {
  const float t = time_it([&] { hipLaunchKernelGGL(fill_passes, dim3(8192), dim3(256), 0, 0, (dbl2*)p, bytes / 16, passes, 1.0); }, 2);
  return (double)bytes * passes / t / 1e9;   // TB/s
}

int main(int argc, char** argv) {
  const size_t chunk_gib = argc > 1 ? atoi(argv[1]) : 2;
  const int nchunks = argc > 2 ? atoi(argv[2]) : 96;
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

  std::vector<hipMemGenericAllocationHandle_t> hs(nchunks);
  std::vector<void*> vas(nchunks, nullptr);
  std::vector<double> rate(nchunks);
  for (int i = 0; i < nchunks; ++i) {
    CK(hipMemCreate(&hs[i], chunk, &prop, 0));
    CK(hipMemAddressReserve(&vas[i], chunk, 0, nullptr, 0));
    CK(hipMemMap(vas[i], chunk, 0, hs[i], 0));
    CK(hipMemSetAccess(vas[i], chunk, &acc, 1));
  }
  for (int round = 0; round < 2; ++round) {
    std::string s = "{\"vmm_chunk_GiB\": " + std::to_string(chunk_gib) + ", \"round\": " + std::to_string(round) + ", \"chunk_TBs\": [";
    for (int i = 0; i < nchunks; ++i) {
      rate[i] = chunk_rate(vas[i], chunk, 8);
      char buf[32];
      snprintf(buf, sizeof buf, "%s%.2f", i ? ", " : "", rate[i]);
      s += buf;
    }
    printf("%s]}\n", s.c_str());
    fflush(stdout);
  }
  std::vector<int> idx(nchunks);
  std::iota(idx.begin(), idx.end(), 0);
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return rate[a] > rate[b]; });
  for (int i = 0; i < nchunks; ++i) CK(hipMemUnmap(vas[i], chunk));
  auto assemble = [&](const char* what, int first) {
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, (size_t)need * chunk, 0, nullptr, 0));
    double mean = 0;
    for (int k = 0; k < need; ++k) {
      CK(hipMemMap((char*)va + (size_t)k * chunk, chunk, 0, hs[idx[first + k]], 0));
      mean += rate[idx[first + k]] / need;
    }
    CK(hipMemSetAccess(va, (size_t)need * chunk, &acc, 1));
    const float seq = time_it([&] { hipLaunchKernelGGL(fill_passes, dim3(8192), dim3(256), 0, 0, (dbl2*)va, out_bytes / 16, 1, 1.0); });
    const float sc = time_it([&] { hipLaunchKernelGGL(scatter_kernel, dim3(7816), dim3(256), 0, 0, (dbl2*)va, order, NROWS, 2.0); });
    printf("{\"assembled\": \"%s\", \"chunks\": %d, \"mean_chunk_TBs\": %.2f, \"seq_ms\": %.3f, \"seq_TBs\": %.2f, "
           "\"scatter_ms\": %.3f, \"scatter_TBs\": %.2f}\n", what, need, mean, seq, out_bytes / seq / 1e9, sc, out_bytes / sc / 1e9);
    fflush(stdout);
    CK(hipMemUnmap(va, (size_t)need * chunk));
    CK(hipMemAddressFree(va, (size_t)need * chunk));
  };
  assemble("fastest", 0);
  assemble("slowest", nchunks - need);
  assemble("fastest-again", 0);
  assemble("middle", (nchunks - need) / 2);
  for (int i = 0; i < nchunks; ++i) {
    CK(hipMemRelease(hs[i]));
    CK(hipMemAddressFree(vas[i], chunk));
  }

  // plain hipMalloc buffers: sustained rate of every 4 GiB piece next to the whole-buffer scatter rate
  const int K = 5;
  std::vector<void*> bufs(K);
  for (int i = 0; i < K; ++i) CK(hipMalloc(&bufs[i], out_bytes));
  for (int i = 0; i < K; ++i) {
    const float sc = time_it([&] { hipLaunchKernelGGL(scatter_kernel, dim3(7816), dim3(256), 0, 0, (dbl2*)bufs[i], order, NROWS, 2.0); });
    std::string s = "{\"hipMalloc\": " + std::to_string(i) + ", \"scatter_ms\": " + std::to_string(sc) + ", \"piece4GiB_TBs\": [";
    const size_t piece = 4ull << 30;
    for (size_t off = 0; off + piece <= out_bytes; off += piece) {
      char buf[32];
      snprintf(buf, sizeof buf, "%s%.2f", off ? ", " : "", chunk_rate((char*)bufs[i] + off, piece, 4));
      s += buf;
    }
    printf("%s]}\n", s.c_str());
    fflush(stdout);
  }
  return 0;
}
