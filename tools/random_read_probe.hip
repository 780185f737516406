// tools/random_read_probe.hip -- the chip's ceiling for the access pattern of the short-channel bilinear
// gather (C5: 8192 x 8192 x 16 f32): per query two naturally aligned 128-byte segments at
// random positions of an 8 GiB table (the pair-packed grid), 64 bytes written.  Measures
//   seg128      random aligned 128-B reads alone, 8 lanes x 16 B per segment (8 full lines per wave instruction)
//   seg128x2    two dependent-address-free segments per query ("row xi" and "row xi+1": the second is a fixed
//               row pitch after the first), still reads only
//   q4          the library kernel's lane mapping: 4 lanes per query, 4 loads of 16 B (two half-lines per load
//               instruction and query), bilinear-like arithmetic, 64-B store per query
//   q8          8 lanes per query: each load instruction covers whole 128-B lines (lanes 0-3 hold the yi
//               corner, lanes 4-7 the yi+1 corner), cross-lane exchange, 64-B store per query
//   seg256/512  256-B and 512-B segments (C3's corner pairs are 512 B)
// with the queries' cell indices pre-generated on the device (uniform).  Prints one JSON object per line.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float flt4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ void gen_kernel(uint32_t* cell, uint64_t n, uint32_t ncells, uint64_t seed) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull;   // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    cell[i] = (uint32_t)(z % ncells);
  }
}

// SEG_LANES lanes x 16 B per segment; reads only, a checksum per thread keeps the loads alive
template <int SEG_LANES, int NSEG, int SHIFT = 0>
__global__ __launch_bounds__(256) void seg_read_kernel(const flt4* tab, const uint32_t* cell, uint64_t nq,
                                                       uint64_t pitch_vecs, float* sink) {
  const uint32_t sub = threadIdx.x % SEG_LANES;
  constexpr uint32_t QPB = 256 / SEG_LANES;
  flt4 acc = {0, 0, 0, 0};
  for (uint64_t q = (uint64_t)blockIdx.x * QPB + threadIdx.x / SEG_LANES; q < nq; q += (uint64_t)gridDim.x * QPB) {
    const uint64_t base = (uint64_t)(cell[q] >> SHIFT) * SEG_LANES;   // same 8 GiB span for every segment size
#pragma unroll
    for (int s = 0; s < NSEG; ++s) acc += tab[base + s * pitch_vecs + sub];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

// the library's mapping (eval_bilinear_kernel, pair-packed grid): 4 lanes per query
__global__ __launch_bounds__(256) void q4_kernel(const flt4* tab, const uint32_t* cell, uint64_t nq,
                                                 uint64_t pitch_vecs, flt4* out) {
  for (uint64_t it = (uint64_t)blockIdx.x * 256 + threadIdx.x; it < nq * 4; it += (uint64_t)gridDim.x * 256) {
    const uint64_t q = it >> 2;
    const uint32_t v = (uint32_t)it & 3u;
    const uint64_t base = (uint64_t)cell[q] * 8;
    const flt4 a11 = tab[base + v], a12 = tab[base + 4 + v];
    const flt4 a21 = tab[base + pitch_vecs + v], a22 = tab[base + pitch_vecs + 4 + v];
    const float tx = 0.25f, ty = 0.75f;
    const flt4 z1 = (a21 - a11) / 1.5f * tx + a11;
    const flt4 z2 = (a22 - a12) / 1.5f * tx + a12;
    out[it] = (z2 - z1) / 2.5f * ty + z1;
  }
}

// 8 lanes per query: every load instruction covers whole lines; lanes 0-3 get the yi+1 corner from lanes 4-7
__global__ __launch_bounds__(256) void q8_kernel(const flt4* tab, const uint32_t* cell, uint64_t nq,
                                                 uint64_t pitch_vecs, flt4* out) {
  for (uint64_t it = (uint64_t)blockIdx.x * 256 + threadIdx.x; it < nq * 8; it += (uint64_t)gridDim.x * 256) {
    const uint64_t q = it >> 3;
    const uint32_t l = (uint32_t)it & 7u;
    const uint64_t base = (uint64_t)cell[q] * 8;
    const flt4 a = tab[base + l];                 // lanes 0-3: z11 channels, lanes 4-7: z12 channels
    const flt4 b = tab[base + pitch_vecs + l];    // z21 | z22
    const float tx = 0.25f, ty = 0.75f;
    const flt4 zx = (b - a) / 1.5f * tx + a;      // z1 on lanes 0-3, z2 on lanes 4-7
    flt4 up;
    up.x = __shfl_down(zx.x, 4);
    up.y = __shfl_down(zx.y, 4);
    up.z = __shfl_down(zx.z, 4);
    up.w = __shfl_down(zx.w, 4);
    if (l < 4) out[q * 4 + l] = (up - zx) / 2.5f * ty + zx;
  }
}

template <class F>
static float time_it(F&& launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  std::vector<float> ts;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

// seg128 reads with the 128-B cells placed `spread` cells apart: the same number of cells and requests over a
// `spread` times larger physical extent (does the random-read rate depend on the extent, like the store rate?)
__global__ __launch_bounds__(256) void spread_read_kernel(const flt4* tab, const uint32_t* cell, uint64_t nq,
                                                          uint64_t spread, uint64_t pitch_vecs, int nseg, float* sink) {
  const uint32_t sub = threadIdx.x & 7u;
  flt4 acc = {0, 0, 0, 0};
  for (uint64_t q = (uint64_t)blockIdx.x * 32 + (threadIdx.x >> 3); q < nq; q += (uint64_t)gridDim.x * 32) {
    const uint64_t base = (uint64_t)cell[q] * 8 * spread;
    acc += tab[base + sub];
    if (nseg > 1) acc += tab[base + pitch_vecs * spread + sub];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

int main(int argc, char** argv) {
  const uint64_t NQ = 12500000;
  if (argc > 1) {
    const uint64_t nx = 8192, ny = 8192, pitch_vecs = (ny - 1) * 8;
    const uint32_t ncells = (uint32_t)((nx - 1) * (ny - 1));
    uint32_t* cell;
    CK(hipMalloc(&cell, NQ * 4));
    hipLaunchKernelGGL(gen_kernel, dim3(4096), dim3(256), 0, 0, cell, NQ, ncells, 12345ull);
    float* sink;
    CK(hipMalloc(&sink, 64));
    for (uint64_t spread : {1ull, 2ull, 4ull, 8ull, 16ull}) {
      const uint64_t bytes = nx * (ny - 1) * 128 * spread + (1 << 20);
      flt4* tab;
      if (hipMalloc(&tab, bytes) != hipSuccess) break;
      CK(hipMemset(tab, 0, bytes));
      for (int nseg = 1; nseg <= 2; ++nseg) {
        float t = time_it([&] { hipLaunchKernelGGL(spread_read_kernel, dim3(8192), dim3(256), 0, 0, tab, cell, NQ, spread, pitch_vecs, nseg, sink); });
        printf("{\"kernel\": \"seg128 spread\", \"spread\": %llu, \"extent_GiB\": %.1f, \"segments_per_query\": %d, \"ms\": %.4f, \"read_TBs\": %.3f}\n",
               (unsigned long long)spread, bytes / 1073741824.0, nseg, t, NQ * 128.0 * nseg / t / 1e9);
      }
      fflush(stdout);
      CK(hipFree(tab));
    }
    return 0;
  }
  const uint64_t nx = 8192, ny = 8192;
  const uint64_t pitch_vecs = (ny - 1) * 8;              // pair-packed row pitch in 16-B vectors (128 B per cell)
  const uint64_t tab_bytes = nx * (ny - 1) * 128;        // 8 GiB
  flt4* tab;
  CK(hipMalloc(&tab, tab_bytes + (1 << 20)));
  CK(hipMemset(tab, 0, tab_bytes + (1 << 20)));
  uint32_t* cell;
  CK(hipMalloc(&cell, NQ * 4));
  const uint32_t ncells = (uint32_t)((nx - 1) * (ny - 1));   // second segment one row pitch further stays inside
  hipLaunchKernelGGL(gen_kernel, dim3(4096), dim3(256), 0, 0, cell, NQ, ncells, 12345ull);
  flt4* out;
  CK(hipMalloc(&out, NQ * 64));
  float* sink;
  CK(hipMalloc(&sink, 64));
  CK(hipDeviceSynchronize());
  const int grids[] = {2048, 8192, 32768};
  for (int g : grids) {
    float t;
    t = time_it([&] { hipLaunchKernelGGL((seg_read_kernel<8, 1>), dim3(g), dim3(256), 0, 0, tab, cell, NQ, pitch_vecs, sink); });
    printf("{\"kernel\": \"seg128\", \"grid\": %d, \"ms\": %.4f, \"Gseg_s\": %.2f, \"read_TBs\": %.3f}\n", g, t, NQ / t / 1e6, NQ * 128.0 / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((seg_read_kernel<8, 2>), dim3(g), dim3(256), 0, 0, tab, cell, NQ, pitch_vecs, sink); });
    printf("{\"kernel\": \"seg128x2\", \"grid\": %d, \"ms\": %.4f, \"Gseg_s\": %.2f, \"read_TBs\": %.3f}\n", g, t, 2 * NQ / t / 1e6, NQ * 256.0 / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL(q4_kernel, dim3(g), dim3(256), 0, 0, tab, cell, NQ, pitch_vecs, out); });
    printf("{\"kernel\": \"q4 (library mapping)\", \"grid\": %d, \"ms\": %.4f, \"Gseg_s\": %.2f, \"read_TBs\": %.3f, \"alg_TBs\": %.3f}\n", g, t, 2 * NQ / t / 1e6, NQ * 256.0 / t / 1e9, NQ * 320.0 / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL(q8_kernel, dim3(g), dim3(256), 0, 0, tab, cell, NQ, pitch_vecs, out); });
    printf("{\"kernel\": \"q8 (whole lines per load)\", \"grid\": %d, \"ms\": %.4f, \"Gseg_s\": %.2f, \"read_TBs\": %.3f, \"alg_TBs\": %.3f}\n", g, t, 2 * NQ / t / 1e6, NQ * 256.0 / t / 1e9, NQ * 320.0 / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((seg_read_kernel<16, 1, 1>), dim3(g), dim3(256), 0, 0, tab, cell, NQ / 2, 0, sink); });
    printf("{\"kernel\": \"seg256\", \"grid\": %d, \"ms\": %.4f, \"read_TBs\": %.3f}\n", g, t, NQ / 2 * 256.0 / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((seg_read_kernel<32, 1, 2>), dim3(g), dim3(256), 0, 0, tab, cell, NQ / 4, 0, sink); });
    printf("{\"kernel\": \"seg512\", \"grid\": %d, \"ms\": %.4f, \"read_TBs\": %.3f}\n", g, t, NQ / 4 * 512.0 / t / 1e9);
    fflush(stdout);
  }
  // the write side alone: 64 B per query, sequential
  float t = time_it([&] { hipLaunchKernelGGL(q4_kernel, dim3(8192), dim3(256), 0, 0, tab, cell, (uint64_t)0, pitch_vecs, out); });
  printf("{\"kernel\": \"empty launch\", \"ms\": %.4f}\n", t);
  return 0;
}
