// Probe (not product code): what does the memory system give the C5 access mix -- random cells of a table far beyond the
// Infinity Cache, 4 x 64 B of corners read and 64 B written per query, four lanes per query -- when a query's corners are
//   (a) two 128-byte aligned records one grid row apart (the pair-packed layout the library ships), or
//   (b) ONE 256-byte aligned record (a cell-quad copy of the grid: 4 x the grid, affordable in 288 GB)?
//   hipcc --offload-arch=gfx950 -O3 -o quad_layout_probe tools/quad_layout_probe.hip && ./quad_layout_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z *= 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
// MODE 0: pair records (128 B) at cell and cell + row; MODE 1: quad record (256 B) at cell
template <int MODE>
__global__ __launch_bounds__(256) void probe(const f4* __restrict__ tab, uint64_t ncx, uint64_t ncy, uint64_t nq, uint64_t seed, f4* __restrict__ out) {
  const uint64_t items = nq * 4;
  for (uint64_t it = (uint64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (uint64_t)gridDim.x * 256) {
    const uint64_t q = it >> 2; const uint32_t v = (uint32_t)(it & 3);
    const uint64_t z = mix(q + seed);
    const uint64_t xi = (z >> 32) % (ncx - 1), yi = (z & 0xffffffffu) % ncy;
    f4 a11, a12, a21, a22;
    if (MODE == 0) {
      const f4* r0 = tab + (xi * ncy + yi) * 8;          // 128-byte record = 8 vectors: [z(yi) 4 vec | z(yi+1) 4 vec]
      const f4* r1 = r0 + ncy * 8;
      a11 = r0[v]; a12 = r0[4 + v]; a21 = r1[v]; a22 = r1[4 + v];
    } else {
      const f4* r = tab + (xi * ncy + yi) * 16;          // 256-byte record
      a11 = r[v]; a12 = r[4 + v]; a21 = r[8 + v]; a22 = r[12 + v];
    }
    const f4 z1 = (a21 - a11) * 0.25f + a11, z2 = (a22 - a12) * 0.25f + a12;
    __builtin_nontemporal_store((z2 - z1) * 0.75f + z1, out + q * 4 + v);
  }
}

int main() {
  const uint64_t nc = 8191, nq = 12500000;
  const size_t pair_bytes = (size_t)(nc + 1) * nc * 128, quad_bytes = (size_t)nc * nc * 256;
  f4 *tab = nullptr, *out = nullptr;
  CK(hipMalloc(&tab, quad_bytes));
  CK(hipMalloc(&out, nq * 64));
  CK(hipMemset(tab, 0, quad_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid_mult : {8, 16, 32}) {
    for (int mode = 0; mode < 2; ++mode) {
      std::vector<float> ms;
      for (int rep = 0; rep < 7; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256 * grid_mult), dim3(256), 0, 0, tab, nc + 1, nc, nq, (uint64_t)rep * 77, out);
        else hipLaunchKernelGGL(probe<1>, dim3(256 * grid_mult), dim3(256), 0, 0, tab, nc + 1, nc, nq, (uint64_t)rep * 77, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      const double bytes = (double)nq * (256 + 64);
      std::printf("{\"layout\": \"%s\", \"wg_per_cu\": %d, \"table_GB\": %.1f, \"median_ms\": %.4f, \"min_ms\": %.4f, \"TBps\": %.2f}\n",
                  mode ? "quad 256 B" : "pair 2 x 128 B", grid_mult, (mode ? quad_bytes : pair_bytes) / 1e9, ms[3], ms[0], bytes / ms[3] / 1e9);
    }
  }
  return 0;
}
