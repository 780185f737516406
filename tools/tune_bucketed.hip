// tools/tune_bucketed.hip -- tuning harness for eval_bucketed_kernel on the Target shape (not part of the product).
// Output goes into a 4-slot row-striped ring (the layout the library recommends), so the numbers do not depend on
// which physical pages a single 32.8 GB buffer got.  Times
//   * the store-only ceiling of the same scattered row stream (scatter), at full and at reduced occupancy,
//   * the library kernel for U in {1,2,4,8} x CQ in {64,128,256}, FULL (straight-line) and bounds-checked,
// on queries grouped by interval on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/tune_bucketed tools/tune_bucketed.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#include "../ndarray-interp_amd/csrc/kernels.hpp"

using namespace ndi;
#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e = (x);                                                              \
    if (e != hipSuccess) {                                                           \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);                \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

template <int CQ>
__global__ __launch_bounds__(256) void scatter_kernel(dbl2* out, const uint32_t* order, uint32_t nrows,
                                                      uint64_t stride_vecs, double v) {
  extern __shared__ unsigned char pad[];   // dynamic LDS only to limit the workgroups per CU
  __shared__ uint32_t s_row[CQ];
  const uint32_t nchunks = (nrows + CQ - 1) / CQ, per = (nchunks + 7) / 8;
  for (uint32_t vb = blockIdx.x; vb < per * 8; vb += gridDim.x) {
    const uint32_t chunk = (vb & 7u) * per + (vb >> 3);
    if (chunk >= nchunks) continue;
    const uint32_t p0 = chunk * CQ, cnt = min((uint32_t)CQ, nrows - p0);
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

// eval_bucketed_kernel<double, ST_CUBIC, 8, 128, true, FULL> with parts switched off, to see what the 6 % over the
// store-only stream is made of.  MODE bit 0: no table reloads (the registers keep the first interval's rows);
// bit 1: no polynomial (the value stored is yl + (1-t)).
template <int MODE, int TB = 256, int U = 8>
__global__ __launch_bounds__(TB) void variant_kernel(Eval1Args<double> A) {
  using V = dbl2;
  constexpr int CQ = 128;
  __shared__ uint32_t s_q[CQ];
  __shared__ uint32_t s_i[CQ];
  __shared__ double s_s[CQ];
  const uint64_t LV = A.lanes / 2;
  const unsigned long long n_valid = A.nq;
  const uint64_t nchunks = (n_valid + CQ - 1) / CQ;
  const uint64_t per = (nchunks + 7) / 8;
  for (uint64_t vb = blockIdx.x; vb < per * 8; vb += gridDim.x) {
    const uint64_t chunk = (vb & 7u) * per + (vb >> 3);
    if ((vb >> 3) >= per || chunk >= nchunks) continue;
    const uint64_t p0 = chunk * CQ;
    const uint32_t cnt = (n_valid - p0 < (uint64_t)CQ) ? (uint32_t)(n_valid - p0) : (uint32_t)CQ;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < cnt; j += TB) {
      const uint4 r = A.rec[p0 + j];
      s_q[j] = r.x;
      s_i[j] = r.y;
      s_s[j] = rec_value(r, 0.0);
    }
    __syncthreads();
    // MODE bit 4: every wave owns a contiguous 8 KiB piece of the row (its 8 stores are adjacent) instead of
    // 1 KiB out of every 4 KiB
    const uint64_t v0 = (MODE & 16) ? (uint64_t)(threadIdx.x >> 6) * (64 * U) + (threadIdx.x & 63) : threadIdx.x;
    constexpr uint64_t VSTEP = (MODE & 16) ? 64 : TB;
    V ryl[U], ryr[U], ra[U], rb[U];
    uint32_t cur = 0xffffffffu;
    if (MODE & 128) {   // four live operand rows that are never loaded from memory
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double f = (double)(threadIdx.x + 256 * u) * 1e-3 + A.out_stride * 1e-9;
        ryl[u] = V{f, f + 1.0}; ryr[u] = V{f * 2.0, f - 1.0}; ra[u] = V{f * 0.5, f * 0.25}; rb[u] = V{f + 3.0, f * 3.0};
      }
      cur = 0;
    }
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_i[j]);
      const uint32_t qi = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_q[j]);
      const double sj = s_s[j];
      if ((MODE & 128) ? false : ((MODE & 1) ? (cur == 0xffffffffu) : (i != cur))) {
        cur = i;
        const V* yl = reinterpret_cast<const V*>(A.data + (uint64_t)i * A.lanes);
        const V* yr = yl + LV;
        const V* pa = reinterpret_cast<const V*>(A.ca + (uint64_t)i * A.lanes);
        const V* pb = reinterpret_cast<const V*>(A.cb + (uint64_t)i * A.lanes);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint64_t v = v0 + (uint64_t)u * VSTEP;
          ryl[u] = yl[v]; ryr[u] = yr[v]; ra[u] = pa[v]; rb[u] = pb[v];
        }
      }
      const RowCoef<double, ST_CUBIC> c = row_coef<double, ST_CUBIC>(A.knots, i, sj, sj);
      V* o = reinterpret_cast<V*>(A.out + (uint64_t)qi * A.out_stride);
      if (MODE & 64) {   // all U results first, then the U stores back to back
        V r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = row_point<double, ST_CUBIC, V>(c, ryl[u], ryr[u], ra[u], rb[u]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) store_stream<true>(o + (v0 + (uint64_t)u * VSTEP), r[u]);
        continue;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t v = v0 + (uint64_t)u * VSTEP;
        if (MODE & 2) store_stream<true>(o + v, ryl[u] + c.c0);
        else if (MODE & 4) store_stream<true>(o + v, c.c0 * ryl[u] + c.c1 * ryl[u] + c.c2 * (ryl[u] * c.c0 + ryl[u] * c.c1));   // 8 operations on ONE table
        else if (MODE & 8) {
#pragma clang fp contract(fast)
          store_stream<true>(o + v, c.c0 * ryl[u] + c.c1 * ryr[u] + c.c2 * (ra[u] * c.c0 + rb[u] * c.c1));   // fused: 5 instructions
        } else store_stream<true>(o + v, row_point<double, ST_CUBIC, V>(c, ryl[u], ryr[u], ra[u], rb[u]));
      }
    }
  }
}

template <class F>
static double time_ms(F&& launch, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  std::vector<double> ts;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const uint64_t n = 4096, L = 4096, Q = 1000000, SLOTS = 4;
  std::mt19937_64 rng(42);
  std::uniform_real_distribution<double> U01(0.0, 1.0);
  std::vector<double> x(2 * n);
  for (auto& v : x) v = U01(rng);
  std::sort(x.begin(), x.end());
  x.resize(n);
  std::vector<double> q(Q), t(Q);
  std::vector<uint32_t> idx(Q), perm(Q);
  for (uint64_t i = 0; i < Q; ++i) {
    q[i] = x[0] + (x[n - 1] - x[0]) * U01(rng);
    uint32_t k = (uint32_t)(std::upper_bound(x.begin(), x.end(), q[i]) - x.begin());
    k = k == 0 ? 0 : k - 1;
    if (k > n - 2) k = n - 2;
    idx[i] = k;
    t[i] = (q[i] - x[k]) / (x[k + 1] - x[k]);
  }
  std::iota(perm.begin(), perm.end(), 0u);
  std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return idx[a] < idx[b]; });
  double *d_x, *d_q, *d_t, *d_data, *d_a, *d_b, *d_out;
  uint32_t *d_idx, *d_perm;
  StatusBlock* d_st;
  CK(hipMalloc(&d_x, n * 8)); CK(hipMemcpy(d_x, x.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_q, Q * 8)); CK(hipMemcpy(d_q, q.data(), Q * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_t, Q * 8)); CK(hipMemcpy(d_t, t.data(), Q * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_idx, Q * 4)); CK(hipMemcpy(d_idx, idx.data(), Q * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_perm, Q * 4)); CK(hipMemcpy(d_perm, perm.data(), Q * 4, hipMemcpyHostToDevice));
  uint4* d_rec;
  {
    std::vector<uint4> rec(Q);
    for (uint64_t p = 0; p < Q; ++p) {
      const uint32_t qi = perm[p];
      unsigned long long b;
      memcpy(&b, &t[qi], 8);
      rec[p] = make_uint4(qi, idx[qi], (uint32_t)b, (uint32_t)(b >> 32));
    }
    CK(hipMalloc(&d_rec, Q * 16)); CK(hipMemcpy(d_rec, rec.data(), Q * 16, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&d_st, sizeof(StatusBlock)));
  CK(hipMemset(d_st, 0xFF, 16));
  const size_t tab = n * L * 8;
  CK(hipMalloc(&d_data, tab)); CK(hipMalloc(&d_a, tab)); CK(hipMalloc(&d_b, tab));
  {
    std::vector<double> h(n * L);
    for (auto& v : h) v = U01(rng);
    CK(hipMemcpy(d_data, h.data(), tab, hipMemcpyHostToDevice));
    std::reverse(h.begin(), h.end());
    CK(hipMemcpy(d_a, h.data(), tab, hipMemcpyHostToDevice));
    std::rotate(h.begin(), h.begin() + 12345, h.end());
    CK(hipMemcpy(d_b, h.data(), tab, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&d_out, SLOTS * Q * L * 8));   // 131 GB: 4 slots interleaved row by row
  const double out_gb = Q * L * 8 / 1e9;
  const uint64_t stride = SLOTS * L;

  // store-only ceiling, full occupancy and limited to 3 / 2 workgroups per CU (dynamic LDS 48 / 72 KiB)
  for (size_t lds : {(size_t)0, (size_t)48 * 1024, (size_t)72 * 1024}) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    double tm = time_ms([&] { hipLaunchKernelGGL(scatter_kernel<128>, dim3(7816), dim3(256), lds, 0, (dbl2*)d_out, d_perm, (uint32_t)Q, stride / 2, 1.0); });
    printf("{\"kernel\": \"scatter store-only\", \"dyn_lds_KiB\": %zu, \"ms\": %.3f, \"TBs\": %.2f}\n", lds / 1024, tm, out_gb / tm);
  }
  Eval1Args<double> A{};
  A.knots = d_x; A.data = d_data; A.ca = d_a; A.cb = d_b; A.q = d_q; A.idx = d_idx; A.t = d_t; A.out = d_out;
  A.lanes = L; A.out_stride = stride; A.nq = Q; A.status = d_st; A.rec = d_rec; A.run = 4;
  const uint64_t LV = L / 2;
#define RUN_BK(U, CQ, FULL)                                                                                   \
  {                                                                                                           \
    const unsigned segs = (unsigned)((LV + 256 * U - 1) / (256 * U));                                         \
    const uint64_t per_xcd = ((Q + CQ - 1) / CQ + 7) / 8;                                                     \
    const unsigned gx = (unsigned)std::min<uint64_t>(per_xcd * 8, 65528);                                     \
    for (int slot = 0; slot < 2; ++slot) {                                                                    \
      A.out = d_out + slot * L;                                                                               \
      double tm = time_ms([&] {                                                                               \
        hipLaunchKernelGGL((eval_bucketed_kernel<double, ST_CUBIC, U, CQ, true, FULL>), dim3(gx, segs), dim3(256), 0, 0, A); \
      });                                                                                                     \
      printf("{\"kernel\": \"bucketed\", \"U\": %d, \"CQ\": %d, \"full\": %d, \"slot\": %d, \"grid\": [%u, %u], \"ms\": %.3f, \"TBs\": %.2f}\n", \
             U, CQ, (int)FULL, slot, gx, segs, tm, out_gb / tm);                                              \
    }                                                                                                         \
    fflush(stdout);                                                                                           \
  }
#define RUN_VAR(MODE)                                                                                         \
  {                                                                                                           \
    A.out = d_out;                                                                                            \
    double tm = time_ms([&] { hipLaunchKernelGGL(variant_kernel<MODE>, dim3(7816), dim3(256), 0, 0, A); });   \
    printf("{\"kernel\": \"variant\", \"no_reloads\": %d, \"no_polynomial\": %d, \"five_ops\": %d, \"fma\": %d, \"ms\": %.3f, \"TBs\": %.2f}\n", MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, (MODE >> 3) & 1, tm, out_gb / tm); \
  }
#define RUN_VAR2(MODE, TB, UU)                                                                                \
  {                                                                                                           \
    A.out = d_out;                                                                                            \
    double tm = time_ms([&] { hipLaunchKernelGGL((variant_kernel<MODE, TB, UU>), dim3(7816), dim3(TB), 0, 0, A); });   \
    printf("{\"kernel\": \"variant\", \"mode\": %d, \"threads\": %d, \"U\": %d, \"ms\": %.3f, \"TBs\": %.2f}\n", MODE, TB, UU, tm, out_gb / tm); \
  }
  RUN_VAR(0) RUN_VAR(1) RUN_VAR(2) RUN_VAR(3) RUN_VAR2(128, 256, 8) RUN_VAR2(0, 256, 8) RUN_VAR2(128, 256, 8)
  if (getenv("TUNE_SWEEP")) {
  RUN_BK(8, 128, true) RUN_BK(8, 128, false) RUN_BK(4, 128, true) RUN_BK(2, 128, true) RUN_BK(1, 128, true)
  RUN_BK(8, 64, true) RUN_BK(8, 256, true) RUN_BK(4, 256, true) RUN_BK(4, 64, true) RUN_BK(8, 512, true)
  }
  return 0;
}
