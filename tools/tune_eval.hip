// tools/tune_eval.hip -- kernel tuning harness (not part of the product): times store/copy ceilings and
// variants of the evaluation kernels of csrc/kernels.hpp on the C2 shape with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/tune_eval tools/tune_eval.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../ndarray-interp_amd/csrc/kernels.hpp"

using namespace ndi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(dbl2* p, size_t nvec, double v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    dbl2 x = {v, v + 1.0};
    store_stream<NT>(p + i, x);
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(dbl2* dst, const dbl2* src, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256)
    store_stream<NT>(dst + i, src[i]);
}

template <class F>
static double time_ms(F&& launch, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  std::vector<double> ts;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  const uint64_t n = 4096, L = 4096, Q = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
  std::mt19937_64 rng(42);
  std::uniform_real_distribution<double> U01(0.0, 1.0);
  std::vector<double> x(2 * n);
  for (auto& v : x) v = U01(rng);
  std::sort(x.begin(), x.end());
  x.resize(n);
  std::vector<double> q(Q);
  for (auto& v : q) v = x[0] + (x[n - 1] - x[0]) * U01(rng);
  // pyramid
  const uint32_t n1 = (n + 63) / 64, n2 = (n1 + 63) / 64;
  std::vector<double> pyr(n + n1 + n2);
  std::copy(x.begin(), x.end(), pyr.begin());
  for (uint32_t j = 0; j < n1; ++j) pyr[n + j] = x[j * 64];
  for (uint32_t j = 0; j < n2; ++j) pyr[n + n1 + j] = x[j * 4096];
  double *d_pyr, *d_q, *d_t, *d_data, *d_a, *d_b, *d_out;
  uint32_t *d_idx, *d_counts, *d_cursor;
  uint4* d_rec;
  StatusBlock* d_st;
  CK(hipMalloc(&d_pyr, pyr.size() * 8)); CK(hipMemcpy(d_pyr, pyr.data(), pyr.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_q, Q * 8)); CK(hipMemcpy(d_q, q.data(), Q * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_t, Q * 8)); CK(hipMalloc(&d_idx, Q * 4)); CK(hipMalloc(&d_rec, Q * 16));
  CK(hipMalloc(&d_counts, n * 4)); CK(hipMalloc(&d_cursor, n * 4)); CK(hipMalloc(&d_st, sizeof(StatusBlock)));
  const size_t tab = n * L * 8;
  CK(hipMalloc(&d_data, tab)); CK(hipMalloc(&d_a, tab)); CK(hipMalloc(&d_b, tab));
  CK(hipMalloc(&d_out, Q * L * 8));
  {  // random tables
    std::vector<double> h(n * L);
    for (auto& v : h) v = U01(rng);
    CK(hipMemcpy(d_data, h.data(), tab, hipMemcpyHostToDevice));
    std::shuffle(h.begin(), h.begin() + 100000, rng);
    CK(hipMemcpy(d_a, h.data(), tab, hipMemcpyHostToDevice));
    std::reverse(h.begin(), h.end());
    CK(hipMemcpy(d_b, h.data(), tab, hipMemcpyHostToDevice));
  }
  CK(hipMemset(d_st, 0xFF, 16)); CK(hipMemset((char*)d_st + 16, 0, sizeof(StatusBlock) - 16));
  LocateArgs<double> LA{};
  LA.pyr = Pyramid<double>{d_pyr, d_pyr + n, (uint32_t)n, n1, 2, 0, 64};
  LA.q = d_q; LA.nq = Q; LA.idx = d_idx; LA.t = d_t; LA.first_fail = &d_st->first_fail[0]; LA.mode = EX_NO; LA.stage_lds = 1;
  const size_t lds = (pyr.size() * 8 + 15) & ~(size_t)15;
  for (unsigned blocks : {256u, 512u, 1024u, 2048u}) {
    LA.slice = ((Q + blocks - 1) / blocks + 255) / 256 * 256;
    const unsigned nblk = (unsigned)((Q + LA.slice - 1) / LA.slice);
    double t_loc = time_ms([&] { hipLaunchKernelGGL((locate_kernel<double, true>), dim3(nblk), dim3(256), lds, 0, LA); });
    printf("locate_kernel %4u blocks            %8.3f ms\n", nblk, t_loc);
  }
  {
    uint32_t* d_hist; CK(hipMalloc(&d_hist, 256 * (n - 1) * 4));
    LA.slice = ((Q + 255) / 256 + 255) / 256 * 256; LA.hist = d_hist; LA.nb = n - 1;
    const unsigned nblk = (unsigned)((Q + LA.slice - 1) / LA.slice);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&locate_kernel<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    double t1 = time_ms([&] { hipLaunchKernelGGL((locate_kernel<double, true>), dim3(nblk), dim3(256), lds + (n - 1) * 4, 0, LA); });
    printf("locate+hist %u blocks               %8.3f ms\n", nblk, t1);
    LA.hist = nullptr;
  }
  const uint32_t nb = n - 1;
  const unsigned g = 2048;
  double t_grp = time_ms([&] {
    CK(hipMemsetAsync(d_counts, 0, nb * 4, 0));
    hipLaunchKernelGGL(bucket_count_kernel, dim3(g), dim3(256), 0, 0, d_idx, Q, d_st, d_counts);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1024), 0, 0, d_counts, nb, d_cursor, d_st);
    hipLaunchKernelGGL(bucket_scatter_kernel<double>, dim3(g), dim3(256), 0, 0, d_idx, (const double*)d_t, Q, d_st, d_cursor, d_rec);
  });
  printf("bucket count+scan+scatter          %8.3f ms\n", t_grp);

  const size_t nvec = Q * L / 2;
  const double out_gb = Q * L * 8 / 1e9;
  for (unsigned blocks : {2048u, 8192u, 65536u}) {
    double t1 = time_ms([&] { hipLaunchKernelGGL(fill_kernel<true>, dim3(blocks), dim3(256), 0, 0, (dbl2*)d_out, nvec, 1.0); });
    double t2 = time_ms([&] { hipLaunchKernelGGL(fill_kernel<false>, dim3(blocks), dim3(256), 0, 0, (dbl2*)d_out, nvec, 1.0); });
    printf("fill %5u blocks  nt %8.3f ms (%6.0f GB/s)   plain %8.3f ms (%6.0f GB/s)\n", blocks, t1, out_gb / t1 * 1e3, t2, out_gb / t2 * 1e3);
  }
  {
    const size_t half = nvec / 2;
    double t1 = time_ms([&] { hipLaunchKernelGGL(copy_kernel<true>, dim3(8192), dim3(256), 0, 0, (dbl2*)d_out, (const dbl2*)d_out + half, half); });
    double t2 = time_ms([&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(8192), dim3(256), 0, 0, (dbl2*)d_out, (const dbl2*)d_out + half, half); });
    printf("copy (r+w %.1f GB)  nt %8.3f ms (%6.0f GB/s)   plain %8.3f ms (%6.0f GB/s)\n", out_gb, t1, out_gb / t1 * 1e3, t2, out_gb / t2 * 1e3);
  }

  Eval1Args<double> A{};
  A.knots = d_pyr; A.data = d_data; A.ca = d_a; A.cb = d_b; A.q = d_q; A.idx = d_idx; A.t = d_t; A.out = d_out;
  A.lanes = L; A.out_stride = L; A.nq = Q; A.status = d_st; A.rec = d_rec; A.run = 1;
  const uint64_t LV = L / 2;
#define RUN_BK(U, CQ, NT)                                                                              \
  {                                                                                                    \
    const unsigned segs = (unsigned)((LV + 256 * U - 1) / (256 * U));                                  \
    const unsigned gx = (unsigned)std::min<uint64_t>((Q + CQ - 1) / CQ, 65535);                        \
    double t = time_ms([&] { hipLaunchKernelGGL((eval_bucketed_kernel<double, ST_CUBIC, U, CQ, NT>), dim3(gx, segs), dim3(256), 0, 0, A); }); \
    printf("bucketed U=%d CQ=%4d nt=%d grid=(%u,%u)  %8.3f ms  (%6.0f GB/s out)\n", U, CQ, (int)NT, gx, segs, t, out_gb / t * 1e3); \
  }
  if (getenv("TUNE_ALL")) {
  RUN_BK(1, 128, true) RUN_BK(2, 128, true) RUN_BK(4, 128, true) RUN_BK(8, 128, true)
  RUN_BK(4, 64, true) RUN_BK(4, 256, true) RUN_BK(4, 512, true) RUN_BK(2, 256, true) RUN_BK(2, 512, true) RUN_BK(8, 256, true)
  RUN_BK(4, 128, false) RUN_BK(4, 256, false) RUN_BK(2, 256, false) RUN_BK(8, 256, false)
  }
#define RUN_RW(U, NT)                                                                                  \
  {                                                                                                    \
    const unsigned segs = (unsigned)((LV + 256 * U - 1) / (256 * U));                                  \
    for (unsigned gx : {4096u, 16384u, 65536u}) {                                                      \
      double t = time_ms([&] { hipLaunchKernelGGL((eval_rows_kernel<double, ST_CUBIC, U, NT>), dim3(gx, segs), dim3(256), 0, 0, A); }, 3); \
      printf("gather rows U=%d nt=%d grid=(%u,%u)  %8.3f ms  (%6.0f GB/s alg)\n", U, (int)NT, gx, segs, t, (Q * L * 40.0 + Q * 8) / 1e9 / t * 1e3); \
    }                                                                                                  \
  }
  if (getenv("TUNE_ALL")) { RUN_RW(1, true) RUN_RW(2, true) RUN_RW(4, true) RUN_RW(8, true) RUN_RW(4, false) }
  return 0;
}
