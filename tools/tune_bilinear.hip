// tools/tune_bilinear.hip -- harness: bilinear evaluation (C3 shape) with exact vs reciprocal-multiply division.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
__global__ void fill_rand(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (h >> 8) * (1.0f / 16777216.0f);
  }
}
__global__ void make_queries(float* qx, float* qy, uint32_t* xi, uint32_t* yi, size_t nq, uint32_t nx, uint32_t ny) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (size_t)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u + 12345u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    unsigned g = h * 3266489917u + 1u; g ^= g >> 16;
    float fx = (h >> 8) * (1.0f / 16777216.0f) * (nx - 1), fy = (g >> 8) * (1.0f / 16777216.0f) * (ny - 1);
    uint32_t ix = min((uint32_t)fx, nx - 2), iy = min((uint32_t)fy, ny - 2);
    qx[i] = fx; qy[i] = fy; xi[i] = ix; yi[i] = iy;
  }
}
int main() {
  const uint32_t nx = 2048, ny = 2048, C = 64; const size_t nq = 10000000;
  float *grid, *qx, *qy, *out, *xk, *yk; uint32_t *xi, *yi; StatusBlock* st;
  CK(hipMalloc(&grid, (size_t)nx * ny * C * 4)); CK(hipMalloc(&qx, nq * 4)); CK(hipMalloc(&qy, nq * 4));
  CK(hipMalloc(&xi, nq * 4)); CK(hipMalloc(&yi, nq * 4)); CK(hipMalloc(&out, nq * C * 4));
  CK(hipMalloc(&xk, nx * 4)); CK(hipMalloc(&yk, ny * 4)); CK(hipMalloc(&st, sizeof(StatusBlock)));
  std::vector<float> ax(nx); for (uint32_t i = 0; i < nx; ++i) ax[i] = (float)i;
  CK(hipMemcpy(xk, ax.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(yk, ax.data(), ny * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, grid, (size_t)nx * ny * C, 7u);
  hipLaunchKernelGGL(make_queries, dim3(4096), dim3(256), 0, 0, qx, qy, xi, yi, nq, nx, ny);
  CK(hipMemset(st, 0xFF, 16)); CK(hipMemset((char*)st + 16, 0, sizeof(StatusBlock) - 16));
  Eval2Args<float> A{}; A.xk = xk; A.yk = yk; A.data = grid; A.qx = qx; A.qy = qy; A.xi = xi; A.yi = yi; A.out = out;
  A.ny = ny; A.row_cells = ny; A.cell_elems = C; A.lanes = C; A.out_stride = C; A.nq = nq; A.status = st;
  const double alg = (double)nq * C * 20 + nq * 8.0;
  for (uint32_t tile_q : {16u, 64u, 256u}) for (unsigned gx : {8192u, 32768u, 131072u}) {
    double t1 = time_ms([&] { hipLaunchKernelGGL((eval_bilinear_kernel<float, 4, false>), dim3(gx), dim3(256), 0, 0, A, tile_q); });
    double t2 = time_ms([&] { hipLaunchKernelGGL((eval_bilinear_kernel<float, 4, true>), dim3(gx), dim3(256), 0, 0, A, tile_q); });
    printf("tile_q=%3u grid=%6u  exact %7.3f ms (%5.0f GB/s alg)   reciprocal-multiply %7.3f ms (%5.0f GB/s alg)\n", tile_q, gx, t1, alg / t1 / 1e6, t2, alg / t2 / 1e6);
  }
  return 0;
}
