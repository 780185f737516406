// tools/placement_probe3.hip -- third placement experiment: how does the sustained streaming-store rate depend on
// the SPAN of addresses written concurrently?
//   window   one window of W bytes at offset 0 of a 65.6 GB hipMalloc buffer, W = 64 MiB ... 64 GiB
//   pair     two 256 MiB windows written concurrently, the second `delta` after the first
//   halves   the bucketed evaluation's scattered row stream (1e6 rows of 32 KiB) into the first / second half of
//            the buffer (two ring slots laid out one after the other) ...
//   striped  ... and with the two slots interleaved row by row (row stride 64 KiB, phase 0 / 1), i.e. every chunk
//            spreads over the whole 65.6 GB
// for two 65.6 GB buffers.  One JSON object per line.
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

// blocks with (blockIdx.x & 1) == 0 fill window a, the others window b (b == a: one window)
__global__ __launch_bounds__(256) void fill2(dbl2* a, dbl2* b, size_t nvec, int passes, double v) {
  dbl2* p = (blockIdx.x & 1) ? b : a;
  const size_t half = gridDim.x / 2, me = blockIdx.x / 2;
  for (int k = 0; k < passes; ++k)
    for (size_t i = me * 256 + threadIdx.x; i < nvec; i += half * 256) {
      dbl2 x = {v + k, v};
      __builtin_nontemporal_store(x, p + i);
    }
}
__global__ __launch_bounds__(256) void fill1(dbl2* p, size_t nvec, int passes, double v) {
  for (int k = 0; k < passes; ++k)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
      dbl2 x = {v + k, v};
      __builtin_nontemporal_store(x, p + i);
    }
}

__global__ __launch_bounds__(256) void scatter_kernel(dbl2* out, const uint32_t* order, uint32_t nrows,
                                                      uint64_t stride_vecs, double v) {
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
      dbl2* o = out + (uint64_t)s_row[j] * stride_vecs;
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

// `placement_probe3 N`: one allocation of N slots; every slot on its own (contiguous) and every phase of the N-way
// row-striped layout.  Without arguments: the window / pair experiments on two 2-slot buffers.
static int striped_only(int nslots, uint32_t* order) {
  const size_t slot = (size_t)NROWS * ROW_VECS * 16;
  char* buf;
  CK(hipMalloc((void**)&buf, slot * nslots));
  auto scat = [&](char* base, uint64_t stride) {
    return time_it([&] { hipLaunchKernelGGL(scatter_kernel, dim3(7816), dim3(256), 0, 0, (dbl2*)base, order, NROWS, stride, 2.0); });
  };
  std::string a = "{\"slots\": " + std::to_string(nslots) + ", \"contiguous_ms\": [", b = "], \"striped_ms\": [";
  for (int k = 0; k < nslots; ++k) {
    char t[32];
    snprintf(t, sizeof t, "%s%.3f", k ? ", " : "", scat(buf + k * slot, ROW_VECS));
    a += t;
  }
  for (int k = 0; k < nslots; ++k) {
    char t[32];
    snprintf(t, sizeof t, "%s%.3f", k ? ", " : "", scat(buf + (size_t)k * ROW_VECS * 16, (uint64_t)nslots * ROW_VECS));
    b += t;
  }
  printf("%s%s]}\n", a.c_str(), b.c_str());
  return 0;
}

int main(int argc, char** argv) {
  const size_t slot = (size_t)NROWS * ROW_VECS * 16;   // 32.768 GB
  const size_t bytes = 2 * slot;
  std::mt19937 rng(7);
  std::vector<uint32_t> h(NROWS);
  std::iota(h.begin(), h.end(), 0u);
  std::shuffle(h.begin(), h.end(), rng);
  uint32_t* order;
  CK(hipMalloc(&order, NROWS * 4));
  CK(hipMemcpy(order, h.data(), NROWS * 4, hipMemcpyHostToDevice));
  if (argc > 1) return striped_only(atoi(argv[1]), order);
  for (int b = 0; b < 2; ++b) {
    char* buf;
    CK(hipMalloc((void**)&buf, bytes));
    {
      std::string s = "{\"buffer\": " + std::to_string(b) + ", \"window_TBs\": {";
      bool first = true;
      for (size_t w = 64ull << 20; w <= bytes; w *= 2) {
        const size_t ww = std::min(w, bytes);
        const int passes = (int)std::max<size_t>(1, (16ull << 30) / ww);
        const float t = time_it([&] { hipLaunchKernelGGL(fill1, dim3(8192), dim3(256), 0, 0, (dbl2*)buf, ww / 16, passes, 1.0); }, 2);
        char tmp[64];
        snprintf(tmp, sizeof tmp, "%s\"%zuMiB\": %.2f", first ? "" : ", ", ww >> 20, (double)ww * passes / t / 1e9);
        s += tmp;
        first = false;
        if (ww == bytes) break;
      }
      printf("%s}}\n", s.c_str());
      fflush(stdout);
    }
    {
      const size_t W = 256ull << 20;
      std::string s = "{\"buffer\": " + std::to_string(b) + ", \"pair_256MiB_TBs_by_delta_GiB\": {";
      const double deltas[] = {0.25, 0.5, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 60};
      bool first = true;
      for (double d : deltas) {
        const size_t off = (size_t)(d * (1ull << 30));
        if (off + W > bytes) continue;
        const float t = time_it([&] { hipLaunchKernelGGL(fill2, dim3(8192), dim3(256), 0, 0, (dbl2*)buf, (dbl2*)(buf + off), W / 16, 32, 1.0); }, 2);
        char tmp[64];
        snprintf(tmp, sizeof tmp, "%s\"%g\": %.2f", first ? "" : ", ", d, 2.0 * W * 32 / t / 1e9);
        s += tmp;
        first = false;
      }
      printf("%s}}\n", s.c_str());
      fflush(stdout);
    }
    auto scat = [&](char* base, uint64_t stride) {
      return time_it([&] { hipLaunchKernelGGL(scatter_kernel, dim3(7816), dim3(256), 0, 0, (dbl2*)base, order, NROWS, stride, 2.0); });
    };
    const float h0 = scat(buf, ROW_VECS), h1 = scat(buf + slot, ROW_VECS);
    const float s0 = scat(buf, 2 * ROW_VECS), s1 = scat(buf + ROW_VECS * 16, 2 * ROW_VECS);
    const float seq = time_it([&] { hipLaunchKernelGGL(fill1, dim3(8192), dim3(256), 0, 0, (dbl2*)buf, bytes / 16, 1, 1.0); });
    printf("{\"buffer\": %d, \"ptr\": \"%p\", \"halves_ms\": [%.3f, %.3f], \"striped_ms\": [%.3f, %.3f], \"seq_whole_TBs\": %.2f}\n",
           b, (void*)buf, h0, h1, s0, s1, bytes / seq / 1e9);
    fflush(stdout);
    // keep the first buffer while the second is allocated so that it lands elsewhere; free at exit
  }
  return 0;
}
