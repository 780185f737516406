// csrc/kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
// Everything here is gather/stream work (no dense contraction, so no MFMA), HBM-bound except where noted:
//   locate_kernel / locate2_kernel
//                          per-query interval search (one / both axes): knot pyramid staged in LDS, top level
//                          held one entry per lane and bisected with cross-lane gathers, lower levels bisected
//                          in LDS; O(1) guess first on evenly spaced axes         (vector_extensions.rs:55-111)
//   eval_rows_kernel       GATHER formulation, long rows: one 256-vector row segment per workgroup pass,
//                          16-byte coalesced loads of the operand rows fused with the polynomial,
//                          non-temporal stores                       (linear.rs:94-96, cubic_spline.rs:818-828)
//   eval_flat_kernel       same arithmetic for short / unaligned rows: one output vector per thread (two-kernel form;
//                          the yardstick of the short-row tests)
//   eval_fused_kernel      short rows (< 256 vectors), QUERY ORDER with the search fused in: 64 queries per wave, one
//                          sequential write stream; tables from L2 (plain / interval-packed: pack_intervals_kernel)
//                          or staged in LDS -- {y, a, b}, or {y, k} with a / b re-formed per item
//   eval_fused_sorted_kernel
//                          the same for rows of 128 B - 1 KiB read from L2: a workgroup round's queries ordered by interval
//                          in LDS first, so that neighbouring items share their operand rows through L1
//   eval_bucketed_short_kernel
//                          short rows grouped by interval: sub-workgroup groups keep the operand vectors in registers
//   range_check_kernel     first failing query of a batch before a query-order launch (first-error semantics)
//   eval_small_kernel / eval_small2d_kernel
//                          <= 16 lanes with host buffers: search + evaluation fused in one launch, one query
//                          per thread (latency path for the reference's own bench shapes)
//   group_offsets_kernel, bucket_scan_kernel, group_scatter_kernel (+ bucket_count/_scatter fallback)
//                          block-local counting sort of the queries by interval (LDS histograms and cursors)
//   eval_bucketed_kernel   BUCKETED formulation: grouped order, operand rows held in registers across a
//                          group, output streamed; XCD-aware chunk order
//   eval_fused2d_kernel    2-D gather order, QUERY ORDER with both searches fused in (axes staged in LDS), one reciprocal
//                          per direction and query in a wave-private LDS strip, row-major coalesced stores
//   eval_bilinear_kernel   2-D gather order, two-kernel form (bilinear.rs:83-97), plain or pair-packed grid
//                          (pack_pairs_kernel): axes too long for LDS, small batches
//   coarse_scatter2d_kernel, scan_bin_totals_kernel, fine_scatter2d_kernel
//                          2-D tile grouping in two levels (tile row, then tile) for large batches: whole-line record
//                          runs instead of one 32-byte sector per record (group_scatter2d_kernel: the one-pass form)
//   slope_pack_kernel, eval_slopes2d_kernel
//                          short 2-D rows (up to 64 bytes) on grids beyond LDS: slope records {z, m} -- linear.rs:33's
//                          division done once per grid point --, one contiguous run per query, an item per lane, loads
//                          issued ahead of the previous batch's stores
//   eval_bilinear_tiles_kernel
//                          2-D tile-grouped order (locate2_kernel's tile histogram + group_scatter2d_kernel): every
//                          tile of grid points staged once in LDS (double-buffered through registers) together with
//                          its x slopes; records decoded once at the hand-over; correctly rounded shared-divisor
//                          divisions (div_shared); bound by tile staging + arithmetic
//   probe_gather_kernel    measurement aid: the 2-D gather's memory access mix alone
//   spline_build_*         batched Thomas solve, one lane of the trailing axes per thread, shared (or
//                          per-lane selected) elimination factors          (cubic_spline.rs:310-368, 409-721)
//   spline_blocked_*       the same system for narrow trailing axes on many knots: both sweeps as blocked first-order
//                          recurrences (the one path that is not bit-identical: a few ulp)
//
// Arithmetic is written in the reference's operation order and the translation unit is
// compiled with -ffp-contract=off, so results are bit-identical to a non-fused CPU evaluation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace ndi {

constexpr unsigned long long NO_FAIL = ~0ull;
constexpr int BLOCK = 256;

enum ExtrapMode : int { EX_NO = 0, EX_YES = 1, EX_PERIODIC = 2 };
enum Strat : int { ST_LINEAR = 0, ST_CUBIC = 1 };

// Device-resident per-evaluation status (one per workspace).
struct StatusBlock {
  unsigned long long first_fail[2];  // lowest failing flat query index per axis (x, y)
  unsigned long long n_valid;        // bucketed path: number of grouped queries
  unsigned long long periodic_mismatch;  // build: lanes with y[0] != y[n-1]
  unsigned int ticket;               // group_offsets_scan_kernel: workgroups that have published their bin totals
  unsigned int reserved;
};

// Opt-in checked build (`make debug` -> libndinterp_hip_dbg.so, -DNDI_BOUNDS): every device-side index that
// addresses memory -- interval index, grouped record position, query index taken from a record, tile-local offset,
// histogram bin -- goes through NDI_CHK(index, limit, code).  A violation is RECORDED (first one wins: code, source
// line, index, limit) in a device word the host reads after every synchronising call, and the index is clamped to 0,
// so the kernel runs on inside its buffers: no trap, no fault, no early return past a barrier -- nothing that could
// hang a wave or take the device down.  GPU AddressSanitizer is not available on this pool; this is the substitute
// ("race detection / sanitizers").  In the normal build NDI_CHK(i, ...) is just (i).
#ifdef NDI_BOUNDS
__device__ unsigned long long g_ndi_bounds[4];   // [0] violations seen, [1] code << 32 | line, [2] index, [3] limit
__device__ __forceinline__ unsigned long long ndi_bounds_fail(int code, int line, unsigned long long idx,
                                                              unsigned long long lim) {
  if (atomicAdd(&g_ndi_bounds[0], 1ull) == 0ull) {
    g_ndi_bounds[1] = ((unsigned long long)code << 32) | (unsigned)line;
    g_ndi_bounds[2] = idx;
    g_ndi_bounds[3] = lim;
  }
  return 0ull;
}
template <class I>
__device__ __forceinline__ I ndi_chk(I idx, unsigned long long lim, int code, int line) {
  return (unsigned long long)idx < lim ? idx : (I)ndi_bounds_fail(code, line, (unsigned long long)idx, lim);
}
#define NDI_CHK(idx, lim, code) ::ndi::ndi_chk((idx), (unsigned long long)(lim), (code), __LINE__)
#else
#define NDI_CHK(idx, lim, code) (idx)
#endif
enum BoundsCode : int {
  BC_INTERVAL = 1,      // interval index i, limit n - 1
  BC_QUERY = 2,         // query index read from a grouped record, limit nq
  BC_POSITION = 3,      // grouped record position, limit nq
  BC_BIN = 4,           // histogram / cursor bin, limit nb
  BC_CELL_X = 5, BC_CELL_Y = 6,   // 2-D cell indices, limits nx - 1 / ny - 1
  BC_TILE = 7,          // tile-local grid point offset, limit (2^ts + 1)^2
  BC_STRIP = 8,         // query slot of a wave's strip, limit 64
};

__global__ __launch_bounds__(64) void reset_status_kernel(StatusBlock* s) {
  s->first_fail[0] = NO_FAIL;
  s->first_fail[1] = NO_FAIL;
  s->n_valid = 0;
  s->periodic_mismatch = 0;
  s->ticket = 0;
  s->reserved = 0;
}

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef float flt4 __attribute__((ext_vector_type(4)));

template <class T, int VEC>
struct VecT;
template <>
struct VecT<double, 2> { using type = dbl2; };
template <>
struct VecT<float, 4> { using type = flt4; };
template <>
struct VecT<double, 1> { using type = double; };
template <>
struct VecT<float, 1> { using type = float; };

template <class T>
struct Wide;
template <>
struct Wide<double> { static constexpr int N = 2; };
template <>
struct Wide<float> { static constexpr int N = 4; };

__device__ __forceinline__ float readlane_t(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double readlane_t(double v, int lane) {
  unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Euclid::rem_euclid for floats: r = a % b; r < 0 ? r + |b| : r   (cubic_spline.rs:808)
__device__ __forceinline__ float rem_euclid_t(float a, float b) {
  float r = fmodf(a, b);
  return (r < 0.0f) ? r + fabsf(b) : r;
}
__device__ __forceinline__ double rem_euclid_t(double a, double b) {
  double r = fmod(a, b);
  return (r < 0.0) ? r + fabs(b) : r;
}

// Arrays that are read-only for the duration of a kernel (the plan arrays of the build) can be read through the
// constant address space: the compiler may
// then use scalar loads for wave-uniform indices and need not order the loads against the kernel's own stores
// (through a plain global pointer it must assume they alias, which pins every such load behind the preceding
// store -- and a load behind a store waits for it: vmcnt is in-order and counts stores on CDNA4).  (Tried for the
// per-query index / t loads of the gather kernel as well: 6 % slower on unsorted queries -- the implicit pacing
// helps the HBM-bound stream -- so evaluation keeps plain loads.)
template <class T>
__device__ __forceinline__ T const_load(const T* p, uint64_t i) {
  typedef const __attribute__((address_space(4))) T* cptr;
  return ((cptr)p)[i];
}

// const_load where the array really is read-only for the whole kernel (CL), a plain load where the same kernel wrote it
// in an earlier phase: loads through the constant address space are invariant to the compiler -- it may move them above
// a barrier, and the scalar cache they may be served from does not see the kernel's own vector stores.
template <bool CL, class T>
__device__ __forceinline__ T ro_load(const T* p, uint64_t i) {
  if constexpr (CL) return const_load(p, i);
  else return p[i];
}

// ---------------------------------------------------------------------------------------------
// locate: knot pyramid  lv0 = knots[n], lv1[j] = knots[j * block]  (block = power of two with 64*block >= n,
// so the top level never has more than 64 entries: one per lane)
// ---------------------------------------------------------------------------------------------
template <class T, class PTR>
struct PyramidT {
  PTR lv0;
  PTR lv1;
  uint32_t n, n1;
  int levels;  // 1 (n <= 64: the knots themselves are the top level) or 2
  int guess;   // axis is close to evenly spaced: try the O(1) index guess first
  uint32_t block;  // knots per top-level entry (power of two)
};
template <class T>
using Pyramid = PyramidT<T, const T*>;                      // global memory
template <class T>
using lds_ptr = const __attribute__((address_space(3))) T*;  // explicit LDS pointers: ds_read, not flat_load
template <class T>
using PyramidLds = PyramidT<T, lds_ptr<T>>;

// Cross-lane gather of a register value (ds_bpermute: the LDS crossbar, no memory access).
__device__ __forceinline__ float lane_gather(float v, uint32_t src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double lane_gather(double v, uint32_t src_lane) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Number of knots <= x, one query per lane, wavefront-cooperative:
//  * top pyramid level (<= 64 entries): the wave holds the level in registers, one entry per lane, and
//    every lane bisects over it with cross-lane gathers (ds_bpermute) -- 64 queries are ranked against
//    the whole level in 7 exchange steps with no memory access at all;
//  * the block of `block` knots selected by the top level is searched by the lane itself with a
//    log2(block)-step branch-free bisection in LDS (the block's first entry is known to be <= x).
// NaN compares false everywhere -> 0.
template <class T, class PTR>
__device__ __forceinline__ uint32_t block_last_le(PTR blk, uint32_t len, uint32_t block, T x) {
  uint32_t lo = 0;  // invariant: blk[lo] <= x
  const uint32_t last = len - 1u;
  for (uint32_t step = block >> 1; step >= 1; step >>= 1) {   // log2(block) steps, wave-uniform trip count
    const uint32_t probe = lo + step;
    const T v = blk[probe < last ? probe : last];  // unconditional (clamped) read: no divergent branch
    lo = (probe <= last && v <= x) ? probe : lo;
  }
  return lo;
}

template <class T, class PTR>
__device__ __forceinline__ uint32_t wave_count_le(const PyramidT<T, PTR>& P, T x, uint32_t lane) {
  // The level is selected with plain branches on the (wave-uniform) level count: a `?:` chain over the
  // struct's pointer members makes the compiler spill the struct to scratch and index it at run time.
  T mine;
  uint32_t last;
  if (P.levels == 2) {
    last = P.n1 - 1u;
    mine = P.lv1[lane < last ? lane : last];
  } else {
    last = P.n - 1u;
    mine = P.lv0[lane < last ? lane : last];
  }
  // lane j holds entry j of the top level
  const bool any = lane_gather(mine, 0u) <= x;
  uint32_t lo = 0;  // invariant (when any): top[lo] <= x
#pragma unroll
  for (uint32_t step = 32; step >= 1; step >>= 1) {
    const uint32_t probe = lo + step;
    const T v = lane_gather(mine, probe < last ? probe : last);
    lo = (probe <= last && v <= x) ? probe : lo;
  }
  if (!any) return 0;
  if (P.levels == 1) return lo + 1u;
  const uint32_t base = lo * P.block;
  const uint32_t len = (P.n - base < P.block) ? P.n - base : P.block;
  return base + block_last_le<T, PTR>(P.lv0 + base, len, P.block, x) + 1u;
}

// Bucket index of an axis (a guess that is right by construction, for any strictly rising axis): M uniform buckets
// over [k0, kn], bucket(x) = min(trunc((x - k0) * scale), M - 1), and lut[b] = number of knots whose bucket is < b
// (M + 1 entries, u16, built on the host with the same arithmetic).  bucket() is monotone, so every knot in a lower
// bucket is < x and every knot in a higher bucket is > x: the number of knots <= x is lut[b] plus the knots <= x
// among the few in bucket b itself -- with M >= 2n that is 0-2 knots for well-spread axes, and a per-lane bisection
// over [lut[b], lut[b+1]) covers clustered axes.  It replaces 7 cross-lane + 6-7 LDS bisection steps by ~4 LDS
// reads (the bisection's random ds_read_b32 are bank-conflict bound: profiles/r02_tuning.md).  The reference does the
// same in spirit: an O(1) guess first, a search only around it (vector_extensions.rs:68-110).
template <class T>
struct BucketIndex {
  const uint16_t* lut;   // global memory, m + 1 entries; nullptr: no bucket index for this axis
  uint32_t m;            // buckets (power of two)
  T scale;               // m / (kn - k0), in T
};
using lds_u16 = const __attribute__((address_space(3))) uint16_t*;

template <class T>
__host__ __device__ __forceinline__ uint32_t bucket_of(T x, T k0, T scale, uint32_t m) {
  const T d = x - k0;
  const T f = d * scale;
  const uint32_t b = (uint32_t)f;   // 0 <= f < ~m for k0 <= x <= kn
  return b < m ? b : m - 1u;
}

// Number of knots <= x through the bucket index; knots and lut staged in LDS.  NaN -> 0.
template <class T>
__device__ __forceinline__ uint32_t lut_count_le(lds_ptr<T> k, uint32_t n, lds_u16 lut, uint32_t m, T scale, T k0,
                                                 T kn, T x) {
  if (!(x >= k0)) return 0u;   // below the axis, or NaN
  if (x >= kn) return n;
  const uint32_t b = bucket_of<T>(x, k0, scale, m);
  uint32_t lo = lut[b];
  uint32_t len = lut[b + 1u] - lo;
  while (len > 0u) {           // count of k[lo .. lo+len) <= x (sorted): usually 0-2 knots
    const uint32_t half = len >> 1;
    const uint32_t mid = lo + half;
    if (k[mid] <= x) {
      lo = mid + 1u;
      len -= half + 1u;
    } else {
      len = half;
    }
  }
  return lo;
}

// The same index for axes whose knots do not fit LDS (more than ~19 000 f64 / 38 000 f32 knots: scalar data on 1e5 - 1e6
// knots is a shape the reference's users have): u32 entries, read from global memory (L2-resident: 8 B per knot).  The
// pyramid search from global memory costs log2(block) = 11-14 DEPENDENT L2 reads per query there; this one costs the two
// adjacent lut reads and then the 1-2 knot reads of the bucket.
template <class T>
struct BucketIndex32 {
  const uint32_t* lut;   // m + 1 entries; nullptr: none
  uint32_t m;
  T scale;
};
template <class T>
__device__ __forceinline__ uint32_t lut32_count_le(const T* k, uint32_t n, const uint32_t* lut, uint32_t m, T scale,
                                                   T k0, T kn, T x) {
  if (!(x >= k0)) return 0u;   // below the axis, or NaN
  if (x >= kn) return n;
  const uint32_t b = bucket_of<T>(x, k0, scale, m);
  uint32_t lo = lut[b];
  uint32_t len = lut[b + 1u] - lo;
  while (len > 0u) {
    const uint32_t half = len >> 1;
    const uint32_t mid = lo + half;
    if (k[mid] <= x) {
      lo = mid + 1u;
      len -= half + 1u;
    } else {
      len = half;
    }
  }
  return lo;
}

// Interval index of one query per lane: the O(1) guess of the reference for evenly spaced axes
// (vector_extensions.rs:68-90: mid = calc_frac((k0,0),(kn,n-1),x), accepted iff k[mid] <= x < k[mid+1]),
// and the cooperative search for the whole wave as soon as one lane's guess is not accepted.  Either way the
// result is the unique i with k[i] <= x < k[i+1], clamped to [0, n-2].  Must be called by all 64 lanes.
template <class T, class PTR>
__device__ __forceinline__ uint32_t locate_index(const PyramidT<T, PTR>& P, T k0, T kn, T x, uint32_t lane) {
  uint32_t gi = 0;
  bool need = true;
  if (P.guess) {
    const T m = (T(P.n - 1u) - T(0)) / (kn - k0) * (x - k0) + T(0);
    gi = (m >= T(0)) ? (uint32_t)(m < T(P.n - 2u) ? m : T(P.n - 2u)) : 0u;   // truncation; NaN -> 0
    const T a = P.lv0[gi], b = P.lv0[gi + 1];
    need = !((a <= x) && (x < b));
  }
  uint32_t i = gi;
  if (__any(need ? 1 : 0)) {
    const uint32_t ub = wave_count_le<T, PTR>(P, x, lane);
    uint32_t s = (ub == 0) ? 0u : ub - 1u;
    if (s > P.n - 2u) s = P.n - 2u;
    if (need) i = s;
  }
  return i;
}

// The same through the bucket index (knots and lut in LDS); no cross-lane step.
template <class T>
__device__ __forceinline__ uint32_t locate_index_lut(const PyramidLds<T>& P, lds_u16 lut, uint32_t m, T scale, T k0,
                                                     T kn, T x) {
  const uint32_t ub = lut_count_le<T>(P.lv0, P.n, lut, m, scale, k0, kn, x);
  uint32_t s = (ub == 0) ? 0u : ub - 1u;
  if (s > P.n - 2u) s = P.n - 2u;
  return s;
}

constexpr int LOCATE_QB = 4;   // batches of 64 queries per wave whose queries are requested together (long slices)

template <class T>
struct LocateArgs {
  Pyramid<T> pyr;          // global-memory pyramid
  const T* q;              // queries
  uint64_t nq;
  uint32_t* idx;           // out: interval index per query (nullable)
  int64_t* idx64;          // out: ndi_get_lower_index_batch result, -1 for NaN (nullable)
  T* t;                    // out (cubic): (x - x_l) / (x_r - x_l) (nullable)
  unsigned long long* first_fail;  // atomicMin target
  int mode;                // ExtrapMode
  int stage_lds;           // copy the pyramid into LDS first
  BucketIndex<T> bx;       // bucket index, staged behind the pyramid when bx.lut != nullptr (needs stage_lds)
  BucketIndex32<T> bx32;   // bucket index read from global memory (axes that are not staged)
  // grouping support (BUCKETED formulation): workgroup b handles the contiguous query slice
  // [b*slice, (b+1)*slice) and, if hist != nullptr, leaves its interval histogram in hist[b][nb]
  uint64_t slice;
  uint32_t* hist;
  uint32_t nb;
};

// Body of locate_kernel for one pyramid address space.
template <class T, class PTR, int QB>
__device__ __forceinline__ void locate_slice(const LocateArgs<T>& A, const PyramidT<T, PTR>& P, uint32_t* s_hist,
                                             lds_u16 lut = nullptr) {
  const uint32_t tid = threadIdx.x;
  const T k0 = P.lv0[0];
  const T kn = P.lv0[P.n - 1];
  const uint32_t lane = tid & 63u;
  const uint64_t q_begin = (uint64_t)blockIdx.x * A.slice;
  uint64_t q_end = q_begin + A.slice;
  if (q_end > A.nq) q_end = A.nq;
  // Software-pipelined: the queries of QB batches are requested together, one round ahead, with unconditional
  // (clamped) loads.  The wave's memory-operation counter is in-order and counts stores: the first use of a freshly
  // loaded query waits for every index the wave has stored before it -- once per QB batches instead of per batch.
  // QB = 4 for slices of >= 16 batches per wave (the grouped forms, the ring's chunks: two-axis search -17 %); QB = 1
  // for the many short slices of a plain search, where a round of look-ahead is a quarter of the slice (+18 % there).
  const uint64_t first = q_begin + (uint64_t)(tid >> 6) * 64u;
  if (q_begin >= q_end) return;              // (no barrier follows in this function)
  const uint64_t q_last = q_end - 1u;        // look-ahead past the slice re-reads its last query (one hot line)
  T xq[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const uint64_t p = first + (uint64_t)j * blockDim.x + lane;
    xq[j] = A.q[p < q_end ? p : q_last];
  }
  for (uint64_t round0 = first; round0 < q_end; round0 += (uint64_t)QB * blockDim.x) {
  T xc[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) xc[j] = xq[j];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const uint64_t p = round0 + (uint64_t)(QB + j) * blockDim.x + lane;
    xq[j] = A.q[p < q_end ? p : q_last];
  }
#pragma unroll 1
  for (int jb = 0; jb < QB; ++jb) {
    const uint64_t base = round0 + (uint64_t)jb * blockDim.x;
    if (base >= q_end) break;
    const uint64_t qi = base + lane;
    const bool active = qi < q_end;
    T x = xc[0];
#pragma unroll
    for (int j = 1; j < QB; ++j)
      if (jb == j) x = xc[j];
    if (!active) x = k0;
    const bool inr = (k0 <= x) && (x <= kn);   // Interp1D::is_in_range, interp1d/mod.rs:384-386
    T xs = x;
    if (A.mode == EX_PERIODIC && !inr) xs = rem_euclid_t(x - k0, kn - k0) + k0;
    // unique i with k[i] <= x < k[i+1], clamped to [0, n-2]  (vector_extensions.rs:61-66, 100-110);
    // all 64 lanes take part (cross-lane exchange inside)
    uint32_t i;
    if constexpr (std::is_same<PTR, lds_ptr<T>>::value) {
      i = lut ? locate_index_lut<T>(P, lut, A.bx.m, A.bx.scale, k0, kn, xs) : locate_index<T, PTR>(P, k0, kn, xs, lane);
    } else if (A.bx32.lut) {
      const uint32_t ub = lut32_count_le<T>(P.lv0, P.n, A.bx32.lut, A.bx32.m, A.bx32.scale, k0, kn, xs);
      i = (ub == 0) ? 0u : ub - 1u;
      if (i > P.n - 2u) i = P.n - 2u;
    } else {
      i = locate_index<T, PTR>(P, k0, kn, xs, lane);
    }
    if (!active) continue;
    const bool isnan_q = !(xs == xs);
    const bool bad = (A.mode == EX_NO) ? !inr : isnan_q;
    if (bad && A.first_fail) atomicMin(A.first_fail, (unsigned long long)qi);
    i = NDI_CHK(i, P.n - 1u, BC_INTERVAL);
    if (A.idx) A.idx[qi] = i;
    if (A.idx64) A.idx64[qi] = isnan_q ? (int64_t)-1 : (int64_t)i;
    if (A.t) {
      const T xl = P.lv0[i], xr = P.lv0[i + 1];
      A.t[qi] = (xs - xl) / (xr - xl);  // cubic_spline.rs:818
    }
    if (s_hist) atomicAdd(&s_hist[NDI_CHK(i, A.nb, BC_BIN)], 1u);
  }
  }
}

// One workgroup (256..1024 threads, chosen by the host so that the LDS footprint still allows a full CU of
// waves) per contiguous slice of queries.  LDS: [pyramid | histogram].
// STAGE: the pyramid is copied into LDS first (compile-time, so the search reads are ds_read).
template <class T, bool STAGE, int QB = LOCATE_QB>
__global__ __launch_bounds__(1024) void locate_kernel(LocateArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (A.nq == 0) return;   // (the clamped query loads below address q[nq - 1])
  const uint32_t tid = threadIdx.x;
  uint32_t* s_hist = nullptr;
  size_t hist_off = 0;
  const uint32_t n = A.pyr.n, n1 = A.pyr.n1;
  if (STAGE) {
    T* s0 = reinterpret_cast<T*>(smem_raw);
    T* s1 = s0 + n;
    for (uint32_t i = tid; i < n; i += blockDim.x) s0[i] = A.pyr.lv0[i];
    for (uint32_t i = tid; i < n1; i += blockDim.x) s1[i] = A.pyr.lv1[i];
    hist_off = ((size_t)(n + n1) * sizeof(T) + 15u) & ~(size_t)15u;
  }
  lds_u16 lut = nullptr;
  if (STAGE && A.bx.lut) {   // [pyramid | lut | histogram]; the lut is copied as 32-bit words (m + 1 entries, padded)
    uint32_t* sl = reinterpret_cast<uint32_t*>(smem_raw + hist_off);
    const uint32_t words = (A.bx.m + 2u) / 2u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(A.bx.lut);
    for (uint32_t i = tid; i < words; i += blockDim.x) sl[i] = src[i];
    lut = (lds_u16)(smem_raw + hist_off);
    hist_off += ((size_t)words * 4u + 15u) & ~(size_t)15u;
  }
  if (A.hist) {
    s_hist = reinterpret_cast<uint32_t*>(smem_raw + hist_off);
    for (uint32_t i = tid; i < A.nb; i += blockDim.x) s_hist[i] = 0u;
  }
  __syncthreads();
  if (STAGE) {
    PyramidLds<T> P;
    P.lv0 = (lds_ptr<T>)(smem_raw);
    P.lv1 = P.lv0 + n;
    P.n = n; P.n1 = n1; P.levels = A.pyr.levels; P.guess = A.pyr.guess; P.block = A.pyr.block;
    locate_slice<T, lds_ptr<T>, QB>(A, P, s_hist, lut);
  } else {
    locate_slice<T, const T*, QB>(A, A.pyr, s_hist);
  }
  if (A.hist) {
    __syncthreads();
    uint32_t* dst = A.hist + (uint64_t)blockIdx.x * A.nb;
    for (uint32_t i = tid; i < A.nb; i += blockDim.x) dst[i] = s_hist[i];
  }
}

// 2-D: both axes in one launch (two independent searches per lane hide each other's LDS latency, one
// staging pass, one launch).  Requires both pyramids to fit LDS; otherwise locate_kernel runs once per axis.
template <class T>
struct Locate2Args {
  Pyramid<T> px, py;
  const T* qx;
  const T* qy;
  uint64_t nq;
  uint32_t* xi;
  uint32_t* yi;                    // nullptr: xi[q] receives the cell word xi | yi << 16
  unsigned long long* first_fail;  // [2]: x, y
  int mode;
  uint64_t slice;
  BucketIndex<T> bx, by;   // bucket indices, staged behind the two pyramids when non-null
  // tile-grouped order (2-D BUCKETED): workgroup b also leaves the histogram of its slice over the tiles of
  // 2^sx x 2^sy cells in hist[b][nb] (LDS atomics, as locate_kernel does for the 1-D intervals)
  uint32_t* hist;
  uint32_t nb, sx, sy, nty;
  // two-level grouping (coarse_scatter2d_kernel): workgroup b also leaves the histogram of its slice over the ntx tile
  // ROWS in chist[b][ntx] -- the row sums of its tile histogram
  uint32_t* chist;
  uint32_t ntx;
  // hist_w != 0: the tile histograms are left in column-block-major order, hist[(bin / hist_w) * slices + b][bin % hist_w]
  // -- the share of the columns one coarse_scatter2d_kernel workgroup sums is then ONE contiguous piece (row-major, its
  // reads are `slices` pieces of hist_w words 4 * nb bytes apart: 65 536 page-crossing touches at C3, 0.26 ms)
  uint32_t hist_w;
};

template <class T, int QB = LOCATE_QB>
__global__ __launch_bounds__(1024) void locate2_kernel(Locate2Args<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (A.nq == 0) return;   // (the clamped query loads below address q[nq - 1])
  const uint32_t tid = threadIdx.x;
  const uint32_t nxa = A.px.n + A.px.n1, nya = A.py.n + A.py.n1;
  T* sx = reinterpret_cast<T*>(smem_raw);
  T* sy = sx + nxa;
  for (uint32_t i = tid; i < nxa; i += blockDim.x) sx[i] = A.px.lv0[i];   // the levels are one allocation
  for (uint32_t i = tid; i < nya; i += blockDim.x) sy[i] = A.py.lv0[i];
  __syncthreads();
  PyramidLds<T> PX, PY;
  PX.lv0 = (lds_ptr<T>)(smem_raw);
  PX.lv1 = PX.lv0 + A.px.n;
  PX.n = A.px.n; PX.n1 = A.px.n1; PX.levels = A.px.levels; PX.guess = A.px.guess; PX.block = A.px.block;
  PY.lv0 = PX.lv0 + nxa;
  PY.lv1 = PY.lv0 + A.py.n;
  PY.n = A.py.n; PY.n1 = A.py.n1; PY.levels = A.py.levels; PY.guess = A.py.guess; PY.block = A.py.block;
  lds_u16 lutx = nullptr, luty = nullptr;
  if (A.bx.lut || A.by.lut) {   // either axis may come without an index (its formula guess is exact)
    const size_t off = ((size_t)(nxa + nya) * sizeof(T) + 15u) & ~(size_t)15u;
    uint32_t* sl = reinterpret_cast<uint32_t*>(smem_raw + off);
    const uint32_t wx = A.bx.lut ? (A.bx.m + 2u) / 2u : 0u, wy = A.by.lut ? (A.by.m + 2u) / 2u : 0u;
    const uint32_t* srcx = reinterpret_cast<const uint32_t*>(A.bx.lut);
    const uint32_t* srcy = reinterpret_cast<const uint32_t*>(A.by.lut);
    for (uint32_t i = tid; i < wx; i += blockDim.x) sl[i] = srcx[i];
    for (uint32_t i = tid; i < wy; i += blockDim.x) sl[wx + i] = srcy[i];
    __syncthreads();
    if (wx) lutx = (lds_u16)(smem_raw + off);
    if (wy) luty = (lds_u16)(smem_raw + off + (size_t)wx * 4u);
  }
  uint32_t* s_hist = nullptr;
  if (A.hist) {   // [x pyramid | y pyramid | x lut | y lut | tile histogram]
    size_t off = ((size_t)(nxa + nya) * sizeof(T) + 15u) & ~(size_t)15u;
    if (A.bx.lut) off += (size_t)((A.bx.m + 2u) / 2u) * 4u;
    if (A.by.lut) off += (size_t)((A.by.m + 2u) / 2u) * 4u;
    off = (off + 15u) & ~(size_t)15u;
    s_hist = reinterpret_cast<uint32_t*>(smem_raw + off);
    for (uint32_t i = tid; i < A.nb; i += blockDim.x) s_hist[i] = 0u;
    __syncthreads();
  }
  const T x0 = PX.lv0[0], xn = PX.lv0[PX.n - 1], y0 = PY.lv0[0], yn = PY.lv0[PY.n - 1];
  const uint32_t lane = tid & 63u;
  const uint64_t q_begin = (uint64_t)blockIdx.x * A.slice;
  uint64_t q_end = q_begin + A.slice;
  if (q_end > A.nq) q_end = A.nq;
  const uint64_t first = q_begin + (uint64_t)(tid >> 6) * 64u;
  // (QB: see locate_slice -- one store drain per QB batches instead of per batch)
  const uint64_t q_last = (q_end > q_begin ? q_end : q_begin + 1u) - 1u;   // look-ahead past the slice: its last query again
  T xq[QB], yq[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const uint64_t p = first + (uint64_t)j * blockDim.x + lane;
    const uint64_t pc = p < q_end ? p : q_last;
    xq[j] = A.qx[pc];
    yq[j] = A.qy[pc];
  }
  for (uint64_t round0 = first; round0 < q_end; round0 += (uint64_t)QB * blockDim.x) {
  T xc[QB], yc[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) { xc[j] = xq[j]; yc[j] = yq[j]; }
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const uint64_t p = round0 + (uint64_t)(QB + j) * blockDim.x + lane;
    const uint64_t pc = p < q_end ? p : q_last;
    xq[j] = A.qx[pc];
    yq[j] = A.qy[pc];
  }
  if (QB > 1 && lutx && luty) {   // (uniform) both axes through their bucket indices: no cross-lane step, so the 2 * QB
    // searches of a round are independent chains of LDS reads -- issued together instead of batch after batch (the counters
    // of the batch-after-batch form at C3: 61 % of the wave cycles waiting, VALU busy 32 of 74 us; 16 waves per CU)
    uint32_t ixv[QB], iyv[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const bool act = round0 + (uint64_t)j * blockDim.x + lane < q_end;
      ixv[j] = locate_index_lut<T>(PX, lutx, A.bx.m, A.bx.scale, x0, xn, act ? xc[j] : x0);
      iyv[j] = locate_index_lut<T>(PY, luty, A.by.m, A.by.scale, y0, yn, act ? yc[j] : y0);
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const uint64_t qi = round0 + (uint64_t)j * blockDim.x + lane;
      if (qi >= q_end) continue;
      const T x = xc[j], y = yc[j];
      const uint32_t ix = ixv[j], iy = iyv[j];
      const bool badx = (A.mode == EX_NO) ? !((x0 <= x) && (x <= xn)) : !(x == x);
      const bool bady = (A.mode == EX_NO) ? !((y0 <= y) && (y <= yn)) : !(y == y);
      if (badx) atomicMin(&A.first_fail[0], (unsigned long long)qi);
      if (bady) atomicMin(&A.first_fail[1], (unsigned long long)qi);
      if (A.yi) {
        A.xi[qi] = NDI_CHK(ix, PX.n - 1u, BC_CELL_X);
        A.yi[qi] = NDI_CHK(iy, PY.n - 1u, BC_CELL_Y);
      } else {
        A.xi[qi] = NDI_CHK(ix, PX.n - 1u, BC_CELL_X) | (NDI_CHK(iy, PY.n - 1u, BC_CELL_Y) << 16);
      }
      if (s_hist) atomicAdd(&s_hist[NDI_CHK((ix >> A.sx) * A.nty + (iy >> A.sy), A.nb, BC_BIN)], 1u);
    }
    continue;
  }
#pragma unroll 1
  for (int jb = 0; jb < QB; ++jb) {
    const uint64_t base = round0 + (uint64_t)jb * blockDim.x;
    if (base >= q_end) break;
    const uint64_t qi = base + lane;
    const bool active = qi < q_end;
    T x = xc[0], y = yc[0];
#pragma unroll
    for (int j = 1; j < QB; ++j)
      if (jb == j) { x = xc[j]; y = yc[j]; }
    if (!active) { x = x0; y = y0; }
    const uint32_t ix = lutx ? locate_index_lut<T>(PX, lutx, A.bx.m, A.bx.scale, x0, xn, x)
                             : locate_index<T, lds_ptr<T>>(PX, x0, xn, x, lane);
    const uint32_t iy = luty ? locate_index_lut<T>(PY, luty, A.by.m, A.by.scale, y0, yn, y)
                             : locate_index<T, lds_ptr<T>>(PY, y0, yn, y, lane);
    if (!active) continue;
    // Interp2D::is_in_x_range / is_in_y_range (interp2d/mod.rs:374-379); NaN handling as in locate_slice
    const bool badx = (A.mode == EX_NO) ? !((x0 <= x) && (x <= xn)) : !(x == x);
    const bool bady = (A.mode == EX_NO) ? !((y0 <= y) && (y <= yn)) : !(y == y);
    if (badx) atomicMin(&A.first_fail[0], (unsigned long long)qi);
    if (bady) atomicMin(&A.first_fail[1], (unsigned long long)qi);
    if (A.yi) {
      A.xi[qi] = NDI_CHK(ix, PX.n - 1u, BC_CELL_X);
      A.yi[qi] = NDI_CHK(iy, PY.n - 1u, BC_CELL_Y);
    } else {   // tile-grouped order with compact records: one cell word per query (both axes < 65536 knots)
      A.xi[qi] = NDI_CHK(ix, PX.n - 1u, BC_CELL_X) | (NDI_CHK(iy, PY.n - 1u, BC_CELL_Y) << 16);
    }
    if (s_hist) atomicAdd(&s_hist[NDI_CHK((ix >> A.sx) * A.nty + (iy >> A.sy), A.nb, BC_BIN)], 1u);
  }
  }
  if (s_hist) {
    __syncthreads();
    if (A.hist_w) {
      for (uint32_t i = tid; i < A.nb; i += blockDim.x) {
        const uint32_t cb = i / A.hist_w, j = i - cb * A.hist_w;
        A.hist[((uint64_t)cb * gridDim.x + blockIdx.x) * A.hist_w + j] = s_hist[i];
      }
    } else {
      uint32_t* dst = A.hist + (uint64_t)blockIdx.x * A.nb;
      for (uint32_t i = tid; i < A.nb; i += blockDim.x) dst[i] = s_hist[i];
    }
    if (A.chist) {   // `parts` neighbouring lanes per tile row (one lane alone walked 128 LDS reads while 7/8 of the workgroup
      // idled: 5 us of the kernel's tail at C3), each a share of the row from a rotated start (banks), combined by shuffles
      const uint32_t parts = A.ntx * 8u <= blockDim.x ? 8u : (A.ntx * 4u <= blockDim.x ? 4u : (A.ntx * 2u <= blockDim.x ? 2u : 1u));
      const uint32_t share = (A.nty + parts - 1u) / parts;
      for (uint32_t c0 = 0; c0 < A.ntx; c0 += blockDim.x / parts) {
        const uint32_t c = c0 + tid / parts, p = tid % parts;
        uint32_t sum = 0;
        if (c < A.ntx) {
          const uint32_t* row = s_hist + c * A.nty;
          const uint32_t f0 = p * share, f1 = (f0 + share < A.nty) ? f0 + share : A.nty;
          const uint32_t len = f1 > f0 ? f1 - f0 : 0u;
          uint32_t f = len ? f0 + c % len : 0u;
          for (uint32_t k = 0; k < len; ++k) {
            sum += row[f];
            f = (f + 1u == f1) ? f0 : f + 1u;
          }
        }
        for (uint32_t d = 1; d < parts; d <<= 1) sum += (uint32_t)__shfl_xor((int)sum, (int)d, 64);
        if (c < A.ntx && p == 0u) A.chist[(uint64_t)blockIdx.x * A.ntx + c] = sum;
      }
    }
  }
}

// Correctly rounded division of many numerators by ONE divisor (the per-query knot spacing shared by all channels
// of a row): r = RN(1 / d) is formed once with the IEEE division, then every quotient costs one multiplication and
// four FMAs -- all available as packed 2 x f32 instructions -- instead of the ~10-instruction IEEE sequence with its
// quarter-rate reciprocal:
//     q0 = RN(n r);  e0 = n - d q0 (exact, FMA);  q1 = RN(q0 + e0 r);  e1 = n - d q1 (exact);  q = RN(q1 + e1 r).
// q0 is within 2^-23 |n / d| of the quotient, so q1 is a faithful rounding of it, and for a faithful q1 and a
// correctly rounded reciprocal the last step yields RN(n / d) exactly (Markstein's theorem; Muller et al., Handbook
// of Floating-Point Arithmetic, "division with an FMA") -- provided no intermediate under- or overflows.  The guard
// keeps divisor and numerators inside an exponent window in which every product and residual is a normal number;
// it is ONE test per vector: the smallest magnitude of the components >= N_LO (an exact zero therefore takes the
// IEEE path: flat data lose the speed-up, never the result) and the sum of the magnitudes <= N_HI (a NaN or an
// infinity makes the sum fail).  Outside the window the lane does the IEEE division.  The explicit FMAs compute
// exact residuals; they are not contractions of the reference's expression, whose operation order (linear.rs:33-35)
// is unchanged: m = RN(n / d), then RN(RN(m (x - x1)) + y1).  Pinned against the IEEE path of the gather kernel on
// all 6.4e8 outputs of C3 (test_full_size_c3_bilinear) and by test_bilinear_tile_grouped_lds (values far outside
// the window included).
template <class T>
struct DivWindow;
template <>
struct DivWindow<float> {
  static constexpr float N_LO = 0x1p-60f, N_HI = 0x1p60f, D_LO = 0x1p-40f, D_HI = 0x1p40f;
};
template <>
struct DivWindow<double> {
  static constexpr double N_LO = 0x1p-500, N_HI = 0x1p500, D_LO = 0x1p-400, D_HI = 0x1p400;
};
template <class T>
struct SharedDivisor {
  T d, r;
  bool ok;
};
template <class T>
__device__ __forceinline__ SharedDivisor<T> shared_divisor(T d) {
  SharedDivisor<T> s;
  s.d = d;
  s.r = T(1) / d;
  s.ok = (d >= DivWindow<T>::D_LO) && (d <= DivWindow<T>::D_HI);
  return s;
}
__device__ __forceinline__ bool nums_in_window(flt4 n) {
  const float lo = fminf(fminf(fabsf(n.x), fabsf(n.y)), fminf(fabsf(n.z), fabsf(n.w)));
  // |x| + |y| and |z| + |w| as VOP3 adds with the magnitudes taken by source modifiers: written out because the compiler
  // pairs the two sums into one packed add, which has no such modifiers, and spends four v_and on the magnitudes first
  // (13 -> 8 instructions for the whole test: it runs once per division vector of the 2-D kernels)
  float s1, s2;
  asm("v_add_f32_e64 %0, |%1|, |%2|" : "=v"(s1) : "v"(n.x), "v"(n.y));
  asm("v_add_f32_e64 %0, |%1|, |%2|" : "=v"(s2) : "v"(n.z), "v"(n.w));
  const float sum = s1 + s2;
  return (lo >= DivWindow<float>::N_LO) & (sum <= DivWindow<float>::N_HI);
}
__device__ __forceinline__ bool nums_in_window(float n) {
  const float a = fabsf(n);
  return (a >= DivWindow<float>::N_LO) & (a <= DivWindow<float>::N_HI);
}
__device__ __forceinline__ bool nums_in_window(double n) {
  const double a = fabs(n);
  return (a >= DivWindow<double>::N_LO) & (a <= DivWindow<double>::N_HI);
}
__device__ __forceinline__ bool nums_in_window(dbl2 n) {
  const double lo = fmin(fabs(n.x), fabs(n.y));
  const double sum = fabs(n.x) + fabs(n.y);
  return (lo >= DivWindow<double>::N_LO) & (sum <= DivWindow<double>::N_HI);
}
template <class T, class V>
__device__ __forceinline__ V div_shared(V n, const SharedDivisor<T>& s) {
  const V d = V(s.d), r = V(s.r);
  const V q0 = n * r;
  const V e0 = __builtin_elementwise_fma(-q0, d, n);
  const V q1 = __builtin_elementwise_fma(e0, r, q0);
  const V e1 = __builtin_elementwise_fma(-q1, d, n);
  V q = __builtin_elementwise_fma(e1, r, q1);
  if (__builtin_expect(!(s.ok & nums_in_window(n)), 0)) q = n / s.d;   // outside the window: the IEEE division
  return q;
}
// The same quotient without the branch: `ok` tells the caller whether the result is the correctly rounded one; when it
// is not (divisor or numerator outside the exponent window) the caller redoes that division with the IEEE instruction
// sequence -- once, behind everything else, so that several independent quotients can be formed in straight-line code.
template <class T>
__device__ __forceinline__ T div_shared_fast(T n, const SharedDivisor<T>& s, bool& ok) {
  // (type-generic FMAs: `__builtin_fma` is the DOUBLE builtin -- with T = float it computed every step in f64 between two
  //  conversions, 100 of the 360 vector instructions of the f32 slope-record kernel's loop, and rounded twice)
  const T q0 = n * s.r;
  const T e0 = __builtin_elementwise_fma(-q0, s.d, n);
  const T q1 = __builtin_elementwise_fma(e0, s.r, q0);
  const T e1 = __builtin_elementwise_fma(-q1, s.d, n);
  ok = s.ok & nums_in_window(n);
  return __builtin_elementwise_fma(e1, s.r, q1);
}
// Linear::calc_frac (linear.rs:29-36) with the divisor's reciprocal shared across the row
template <class T, class V>
__device__ __forceinline__ V frac_shared(T x1, V y1, const SharedDivisor<T>& dx, V y2, T x) {
  const V m = div_shared<T, V>(y2 - y1, dx);
  return m * (x - x1) + y1;
}

// ---------------------------------------------------------------------------------------------
// 1-D evaluation
// ---------------------------------------------------------------------------------------------
template <class T>
struct Eval1Args {
  const T* knots;
  const T* data;   // [n][lanes]
  const T* ca;     // [n-1][lanes] (cubic)
  const T* cb;
  const T* q;      // raw queries (linear uses x itself)
  const uint32_t* idx;
  const T* t;      // cubic
  T* out;
  uint64_t lanes, out_stride, nq;
  const StatusBlock* status;
  // bucketed: per grouped position one record {query index, interval, s as raw bits} with s = t (cubic) or the raw
  // query value (linear) -- written by the grouping kernels, read sequentially by the evaluation (the per-query
  // idx[qi] / t[qi] gathers it replaces were 0.13-0.25 GB of random line reads per 1e6 queries, and reads cost the
  // write stream several times their share of the bytes: profiles/r02_tuning.md)
  const uint4* rec;
  uint32_t run;   // consecutive chunks per workgroup (0 = 1)
  uint32_t n_int; // n - 1: limit of every interval index (checked build)
};

template <class T>
__device__ __forceinline__ uint4 make_rec(uint32_t qi, uint32_t i, T s);
template <>
__device__ __forceinline__ uint4 make_rec<double>(uint32_t qi, uint32_t i, double s) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, s);
  return make_uint4(qi, i, (uint32_t)b, (uint32_t)(b >> 32));
}
template <>
__device__ __forceinline__ uint4 make_rec<float>(uint32_t qi, uint32_t i, float s) {
  return make_uint4(qi, i, __builtin_bit_cast(uint32_t, s), 0u);
}
__device__ __forceinline__ double rec_value(const uint4& r, double) {
  return __builtin_bit_cast(double, ((unsigned long long)r.w << 32) | r.z);
}
__device__ __forceinline__ float rec_value(const uint4& r, float) { return __builtin_bit_cast(float, r.z); }

// Per-query scalars shared by all lanes of a row.
template <class T, int STRAT>
struct RowCoef {
  T c0, c1, c2;  // cubic: (1-t), t, t(1-t)      linear: (x2-x1), (x-x1), unused
};

template <class T, int STRAT>
__device__ __forceinline__ RowCoef<T, STRAT> row_coef(const T* knots, uint32_t i, T xq, T t) {
  RowCoef<T, STRAT> c;
  if (STRAT == ST_CUBIC) {
    const T one = T(1);
    c.c0 = one - t;
    c.c1 = t;
    c.c2 = t * (one - t);
  } else {
    const T x1 = knots[i], x2 = knots[i + 1];
    c.c0 = x2 - x1;
    c.c1 = xq - x1;
    c.c2 = T(0);
  }
  return c;
}

// cubic_spline.rs:825-827:  (1-t)*yl + t*yr + t*(1-t)*(a*(1-t) + b*t)
// linear.rs:33-35:          ((y2-y1)/(x2-x1)) * (x-x1) + y1
template <class T, int STRAT, class V>
__device__ __forceinline__ V row_point(const RowCoef<T, STRAT>& c, V yl, V yr, V a, V b) {
  if (STRAT == ST_CUBIC) {
    return c.c0 * yl + c.c1 * yr + c.c2 * (a * c.c0 + b * c.c1);
  } else {
    V m = (yr - yl) / c.c0;
    return m * c.c1 + yl;
  }
}

// Output rows are written once and never re-read by the kernel: non-temporal by default.
template <bool NT, class V>
__device__ __forceinline__ void store_stream(V* p, V v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// Short trailing axes, latency-sensitive batches: search and evaluation fused in one launch, one query per
// thread (lanes <= SMALL_LANES).  Because rows are written before the batch's first failing query is known,
// the host only uses it with a staging buffer it owns (host-output mode) and copies out the rows before the
// failure -- the caller-visible semantics stay those of the two-kernel path.
constexpr int SMALL_LANES = 16;

template <class T>
struct EvalSmallArgs {
  Pyramid<T> pyr;
  const T* data;
  const T* ca;
  const T* cb;
  const T* q;
  T* out;          // [nq][out_stride]
  uint64_t nq, out_stride;
  uint32_t lanes;
  int mode;        // ExtrapMode
  unsigned long long* first_fail;
  int prechecked;  // first_fail already holds the batch's first failing query (range_check_kernel ran):
                   // rows at or after it are skipped, so a caller-owned output keeps them untouched
};

// Range / NaN test alone (the failure conditions of locate_slice), for the fused small-lanes kernels when
// they write straight into a caller-owned device buffer: the first failing index must be known before any
// row is written.  Reads 1 (2) values per query.
template <class T>
__global__ __launch_bounds__(BLOCK) void range_check_kernel(const T* qx, const T* qy, uint64_t nq, T x0, T xn,
                                                            T y0, T yn, int mode, unsigned long long* first_fail) {
  // four independent loads per thread and trip: the pass is a pure read stream and needs the bytes in flight
  constexpr int RU = 4;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  for (uint64_t q0 = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; q0 < nq; q0 += step * RU) {
    T xv[RU], yv[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const uint64_t qi = q0 + (uint64_t)u * step;
      xv[u] = (qi < nq) ? qx[qi] : x0;
      yv[u] = (qy && qi < nq) ? qy[qi] : y0;
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const uint64_t qi = q0 + (uint64_t)u * step;
      if (qi >= nq) break;
      const T x = xv[u];
      const bool inr = (x0 <= x) && (x <= xn);
      T xs = x;
      if (mode == EX_PERIODIC && !inr) xs = rem_euclid_t(x - x0, xn - x0) + x0;   // +-inf wraps to NaN, as in locate_slice
      const bool badx = (mode == EX_NO) ? !inr : !(xs == xs);
      if (badx) atomicMin(&first_fail[0], (unsigned long long)qi);
      if (qy) {
        const T y = yv[u];
        const bool bady = (mode == EX_NO) ? !((y0 <= y) && (y <= yn)) : !(y == y);
        if (bady) atomicMin(&first_fail[1], (unsigned long long)qi);
      }
    }
  }
}

// Interval-packed copy of the tables for rows shorter than a cache line: P[i] = { y[i], y[i+1], a[i], b[i] } (cubic) or
// { y[i], y[i+1] } (linear), i < n - 1.  A query's operands are then ONE contiguous record (2 lines at 8 f64 lanes)
// instead of three separate row pieces that each cost a whole 128-byte line fill (3.5 lines on average): the
// query-order kernel is bound by L2 -> L1 line fills at these shapes (profiles/r04_short_rows.md).  Values are
// copied, never recomputed.  One thread per element of the packed array.
template <class T>
__global__ __launch_bounds__(BLOCK) void pack_intervals_kernel(const T* data, const T* ca, const T* cb, T* out,
                                                               uint64_t n, uint64_t lanes, int parts) {
  const uint64_t total = (n - 1) * (uint64_t)parts * lanes;
  for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (uint64_t)gridDim.x * BLOCK) {
    const uint64_t l = e % lanes;
    const uint64_t r = e / lanes;
    const uint64_t k = r % (uint64_t)parts, i = r / (uint64_t)parts;
    T v;
    if (k == 0) v = data[i * lanes + l];
    else if (k == 1) v = data[(i + 1) * lanes + l];
    else if (k == 2) v = ca[i * lanes + l];
    else v = cb[i * lanes + l];
    out[e] = v;
  }
}

template <class T, int STRAT>
__global__ __launch_bounds__(BLOCK) void eval_small_kernel(EvalSmallArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const uint32_t tid = threadIdx.x;
  const uint32_t n = A.pyr.n, n1 = A.pyr.n1;
  {
    T* s0 = reinterpret_cast<T*>(smem_raw);
    const uint32_t total = n + n1;   // the levels are one allocation
    for (uint32_t i = tid; i < total; i += BLOCK) s0[i] = A.pyr.lv0[i];
  }
  __syncthreads();
  PyramidLds<T> P;
  P.lv0 = (lds_ptr<T>)(smem_raw);
  P.lv1 = P.lv0 + n;
  P.n = n; P.n1 = n1; P.levels = A.pyr.levels; P.guess = A.pyr.guess; P.block = A.pyr.block;
  const T k0 = P.lv0[0], kn = P.lv0[n - 1];
  const uint32_t lane = tid & 63u;
  const uint32_t L = A.lanes;
  unsigned long long limit = A.prechecked ? *A.first_fail : NO_FAIL;
  if (limit > A.nq) limit = A.nq;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK + (tid & ~63u); base < limit; base += (uint64_t)gridDim.x * BLOCK) {
    const uint64_t qi = base + lane;
    const bool active = qi < limit;
    const T x = active ? A.q[qi] : k0;
    const bool inr = (k0 <= x) && (x <= kn);
    T xs = x;
    if (A.mode == EX_PERIODIC && !inr) xs = rem_euclid_t(x - k0, kn - k0) + k0;
    const uint32_t i = locate_index<T, lds_ptr<T>>(P, k0, kn, xs, lane);   // all 64 lanes take part
    if (!active) continue;
    const bool bad = (A.mode == EX_NO) ? !inr : !(xs == xs);
    if (bad) {
      if (!A.prechecked) atomicMin(A.first_fail, (unsigned long long)qi);
      continue;
    }
    const T xl = P.lv0[i], xr = P.lv0[i + 1];
    RowCoef<T, STRAT> c;
    if (STRAT == ST_CUBIC) {
      const T t = (xs - xl) / (xr - xl);   // cubic_spline.rs:818
      const T one = T(1);
      c.c0 = one - t;
      c.c1 = t;
      c.c2 = t * (one - t);
    } else {
      c.c0 = xr - xl;
      c.c1 = x - xl;
      c.c2 = T(0);
    }
    const T* yl = A.data + (uint64_t)i * L;
    const T* yr = yl + L;
    const T* pa = A.ca + (uint64_t)i * L;
    const T* pb = A.cb + (uint64_t)i * L;
    T* o = A.out + qi * A.out_stride;
    for (uint32_t l = 0; l < L; ++l) {
      const T a = (STRAT == ST_CUBIC) ? pa[l] : T(0);
      const T b = (STRAT == ST_CUBIC) ? pb[l] : T(0);
      o[l] = row_point<T, STRAT, T>(c, yl[l], yr[l], a, b);
    }
  }
}

// GATHER, long rows: grid.x strides over queries, grid.y over 256*U-vector segments of a row.
template <class T, int STRAT, int U, bool NT = true>
__global__ __launch_bounds__(BLOCK) void eval_rows_kernel(Eval1Args<T> A) {
  constexpr int VN = Wide<T>::N;
  using V = typename VecT<T, VN>::type;
  const uint64_t LV = A.lanes / VN;
  const uint32_t seg_vecs = BLOCK * U;
  const uint32_t segs = (uint32_t)((LV + seg_vecs - 1) / seg_vecs);
  unsigned long long limit = A.status->first_fail[0];
  if (limit > A.nq) limit = A.nq;
  for (uint64_t qi = blockIdx.x; qi < limit; qi += gridDim.x) {
    const uint32_t i = NDI_CHK(A.idx[qi], A.n_int, BC_INTERVAL);
    const RowCoef<T, STRAT> c =
        row_coef<T, STRAT>(A.knots, i, STRAT == ST_LINEAR ? A.q[qi] : T(0),
                           STRAT == ST_CUBIC ? A.t[qi] : T(0));
    const V* yl = reinterpret_cast<const V*>(A.data + (uint64_t)i * A.lanes);
    const V* yr = yl + LV;
    const V* pa = reinterpret_cast<const V*>(A.ca + (uint64_t)i * A.lanes);
    const V* pb = reinterpret_cast<const V*>(A.cb + (uint64_t)i * A.lanes);
    V* o = reinterpret_cast<V*>(A.out + qi * A.out_stride);
    for (uint32_t seg = blockIdx.y; seg < segs; seg += gridDim.y) {
      const uint64_t v0 = (uint64_t)seg * seg_vecs + threadIdx.x;
      V ryl[U], ryr[U], ra[U], rb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t v = v0 + (uint64_t)u * BLOCK;
        if (v < LV) {
          ryl[u] = yl[v];
          ryr[u] = yr[v];
          if (STRAT == ST_CUBIC) {
            ra[u] = pa[v];
            rb[u] = pb[v];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t v = v0 + (uint64_t)u * BLOCK;
        if (v < LV) store_stream<NT>(o + v, row_point<T, STRAT, V>(c, ryl[u], ryr[u], ra[u], rb[u]));
      }
    }
  }
}

// GATHER, short / unaligned rows: one VEC-wide output vector per thread, several queries per
// workgroup tile.  tile_q queries x LV vectors <= 2^31 items per tile.
template <class T, int STRAT, int VEC>
__global__ __launch_bounds__(BLOCK) void eval_flat_kernel(Eval1Args<T> A, uint32_t tile_q) {
  using V = typename VecT<T, VEC>::type;
  const uint32_t LV = (uint32_t)(A.lanes / VEC);
  unsigned long long limit = A.status->first_fail[0];
  if (limit > A.nq) limit = A.nq;
  const uint64_t ntiles = (limit + tile_q - 1) / tile_q;
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t q0 = tile * tile_q;
    const uint32_t nq_here = (limit - q0 < tile_q) ? (uint32_t)(limit - q0) : tile_q;
    const uint32_t items = nq_here * LV;
    for (uint32_t it = threadIdx.x; it < items; it += BLOCK) {
      const uint32_t ql = it / LV;
      const uint32_t v = it - ql * LV;
      const uint64_t qi = q0 + ql;
      const uint32_t i = NDI_CHK(A.idx[qi], A.n_int, BC_INTERVAL);
      const RowCoef<T, STRAT> c =
          row_coef<T, STRAT>(A.knots, i, STRAT == ST_LINEAR ? A.q[qi] : T(0),
                             STRAT == ST_CUBIC ? A.t[qi] : T(0));
      const V* yl = reinterpret_cast<const V*>(A.data + (uint64_t)i * A.lanes);
      const V* yr = reinterpret_cast<const V*>(A.data + (uint64_t)(i + 1) * A.lanes);
      V a = V(0), b = V(0);
      if (STRAT == ST_CUBIC) {
        a = reinterpret_cast<const V*>(A.ca + (uint64_t)i * A.lanes)[v];
        b = reinterpret_cast<const V*>(A.cb + (uint64_t)i * A.lanes)[v];
      }
      V* o = reinterpret_cast<V*>(A.out + qi * A.out_stride);
      o[v] = row_point<T, STRAT, V>(c, yl[v], yr[v], a, b);
    }
  }
}

// GATHER, short rows (fewer than 256 vectors), QUERY ORDER, search fused in: the formulation for the reference's own
// data shapes (scalar data, (100, 5), a few dozen lanes -- benches/bench_interp1d.rs:82-122).  A wave takes 64
// consecutive queries, one per lane: search (bucket index or pyramid, knots staged in LDS) and the per-query scalars
// (cubic_spline.rs:818 / linear.rs:33-35), parked in a wave-private LDS strip; then the wave walks the batch's
// 64 * LV output vectors in row-major order, 64 per trip -- lane -> (query, vector) by one v_mul_hi (LV is not a
// power of two for the reference's 5-lane rows) -- so a trip's stores cover 1 KiB of CONSECUTIVE output bytes and the
// whole batch is one sequential write stream, with no idx[] / t[] round trip through memory (12 B written + 12-20 B
// re-read per query by the two-kernel form: a third of the traffic at 8 f64 lanes).  UNR trips are issued together:
// scalars, then all 4 * UNR operand loads, then arithmetic and stores.
// TLDS == 1: the tables themselves (data, a, b) are staged in LDS once per workgroup -- when they fit (the reference's
// bench shapes do many times over) the operand gathers never leave the CU.
// TLDS == 2 (CubicSpline): {data, k} are staged instead -- two thirds of the bytes, so table sets up to 1.5 x larger
// fit (8 f64 lanes on 1024 knots: 128 KiB) and smaller ones leave room for a second workgroup -- and every item
// re-forms a = k[i] dx - dy, b = dy - k[i+1] dx (cubic_spline.rs:354-365) with the build's own operations in the
// build's order (no contraction: -ffp-contract=off), i.e. the very bits the a / b tables hold.
// The first failing query of the batch is known before the launch (range_check_kernel): rows at / after it are
// never written, as in the reference's serial loop (interp1d/mod.rs:334-342).
template <class T>
struct EvalFusedArgs {
  Pyramid<T> pyr;
  BucketIndex<T> bx;     // lut == nullptr: pyramid search
  const T* data;
  const T* ca;
  const T* cb;
  const T* ck;           // the derivatives k, [n][lanes] (TLDS == 2 only)
  BucketIndex32<T> bx32; // TLDS == 3: bucket index of the axis in global memory (nullptr: pyramid search from memory)
  const T* q;
  T* out;
  uint64_t nq, out_stride;
  uint32_t lanes;
  uint32_t lv;           // vectors per row
  uint32_t lv_magic;     // ceil(2^32 / lv) for lv >= 2: it / lv == mulhi(it, lv_magic) for it * lv < 2^32
  uint32_t rec_stride;   // vectors between the operands of interval i and i + 1: lv for the plain tables; 4 * lv
                         // (cubic) / 2 * lv (linear) for the interval-packed copy (pack_intervals_kernel), where data
                         // / ca / cb point at the y / a / b parts of interval 0's record
  int mode;              // ExtrapMode
  unsigned long long* first_fail;
  int check;             // 1: no range pre-pass ran (NDI_EVAL_FRESH_OUTPUT): the kernel tests every query itself, records
                         // the lowest failing index and writes every row (the output is the call's own, dropped on Err)
  int debug;             // NDI_TUNING builds only (measurement aid, results meaningless): bit 0 no search, bit 1 every
                         // item reads interval (lane & 7), bit 2 no stores, bit 3 no operand loads at all
};

template <class V, bool LDS>
struct TabPtr { using type = const V*; };
template <class V>
struct TabPtr<V, true> { using type = const __attribute__((address_space(3))) V*; };

template <class T>
__device__ __forceinline__ bool lane_query_fails(T x, T k0, T kn, int mode);

template <class T, int STRAT, int VEC, int UNR, int TB, int TLDS>
__global__ __launch_bounds__(TB) void eval_fused_kernel(EvalFusedArgs<T> A) {
  static_assert(TLDS != 2 || STRAT == ST_CUBIC, "the {y, k} form is the spline's");
  // TLDS == 3: axes too long for LDS -- the knots stay in global memory (L2) and are searched through the u32 bucket
  // index (lut32_count_le: two adjacent index reads + the bucket's 1-2 knots); tables from memory as with TLDS == 0.
  // Still one launch and no idx[] / t[] round trip (12 B written + 12-20 B re-read per query by the two-kernel form).
  constexpr bool GK = TLDS == 3;
  constexpr bool TAB_LDS = TLDS == 1 || TLDS == 2;
  using V = typename VecT<T, VEC>::type;
  using tab_ptr = typename TabPtr<V, TAB_LDS>::type;
  constexpr bool STRIP2 = STRAT == ST_LINEAR || TLDS == 2;   // a second per-query scalar: (x - x1) / the interval's dx
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t WAVES = TB / 64;
  if (A.nq == 0) return;   // (the clamped query loads below address q[nq - 1])
  const uint32_t tid = threadIdx.x;
  const uint32_t n = A.pyr.n, n1 = A.pyr.n1;
  // LDS: [pyramid | lut | per-wave strips (interval, c0, c1) | tables]
  size_t off = 0;
  if constexpr (!GK) {
    T* s0 = reinterpret_cast<T*>(smem_raw);
    const uint32_t total = n + n1;   // the levels are one allocation
    for (uint32_t i = tid; i < total; i += TB) s0[i] = A.pyr.lv0[i];
    off = ((size_t)total * sizeof(T) + 15u) & ~(size_t)15u;
  }
  lds_u16 lut = nullptr;
  if (!GK && A.bx.lut) {
    uint32_t* sl = reinterpret_cast<uint32_t*>(smem_raw + off);
    const uint32_t words = (A.bx.m + 2u) / 2u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(A.bx.lut);
    for (uint32_t i = tid; i < words; i += TB) sl[i] = src[i];
    lut = (lds_u16)(smem_raw + off);
    off += ((size_t)words * 4u + 15u) & ~(size_t)15u;
  }
  uint32_t* w_i = reinterpret_cast<uint32_t*>(smem_raw + off) + (tid >> 6) * 64u;
  off += (size_t)WAVES * 64u * sizeof(uint32_t);
  T* w_c0 = reinterpret_cast<T*>(smem_raw + off) + (tid >> 6) * 64u;
  off += (size_t)WAVES * 64u * sizeof(T);
  T* w_c1 = reinterpret_cast<T*>(smem_raw + off) + (tid >> 6) * 64u;
  if (STRIP2) off += (size_t)WAVES * 64u * sizeof(T);
  const uint32_t LV = A.lv;
  const uint32_t RS = TAB_LDS ? LV : A.rec_stride;
  tab_ptr t_y, t_a, t_b;
  if constexpr (TLDS == 2) {
    V* sy = reinterpret_cast<V*>(smem_raw + off);
    const uint32_t ny = n * LV;
    V* sk = sy + ny;
    const V* gy = reinterpret_cast<const V*>(A.data);
    const V* gk = reinterpret_cast<const V*>(A.ck);
    for (uint32_t i = tid; i < ny; i += TB) sy[i] = gy[i];
    for (uint32_t i = tid; i < ny; i += TB) sk[i] = gk[i];
    t_y = (tab_ptr)(smem_raw + off);
    t_a = t_y + ny;          // k
    t_b = t_a;
  } else if constexpr (TLDS == 1) {
    V* sy = reinterpret_cast<V*>(smem_raw + off);
    const uint32_t ny = n * LV, nab = (STRAT == ST_CUBIC) ? (n - 1u) * LV : 0u;
    V* sa = sy + ny;
    V* sb = sa + nab;
    const V* gy = reinterpret_cast<const V*>(A.data);
    const V* ga = reinterpret_cast<const V*>(A.ca);
    const V* gb = reinterpret_cast<const V*>(A.cb);
    for (uint32_t i = tid; i < ny; i += TB) sy[i] = gy[i];
    for (uint32_t i = tid; i < nab; i += TB) sa[i] = ga[i];
    for (uint32_t i = tid; i < nab; i += TB) sb[i] = gb[i];
    t_y = (tab_ptr)(smem_raw + off);
    t_a = t_y + ny;
    t_b = t_a + nab;
  } else {
    t_y = reinterpret_cast<const V*>(A.data);
    t_a = reinterpret_cast<const V*>(A.ca);
    t_b = reinterpret_cast<const V*>(A.cb);
  }
  __syncthreads();
  PyramidLds<T> P;
  P.lv0 = (lds_ptr<T>)(smem_raw);
  P.lv1 = P.lv0 + n;
  P.n = n; P.n1 = n1; P.levels = A.pyr.levels; P.guess = A.pyr.guess; P.block = A.pyr.block;
  const T k0 = GK ? A.pyr.lv0[0] : (T)P.lv0[0], kn = GK ? A.pyr.lv0[n - 1] : (T)P.lv0[n - 1];
  const uint32_t lane = tid & 63u;
  const bool contig = A.out_stride == (uint64_t)A.lanes;
  unsigned long long limit = A.check ? NO_FAIL : *A.first_fail;
  if (limit > A.nq) limit = A.nq;
  const uint64_t wave_step = (uint64_t)gridDim.x * TB;
  // The queries of QB batches are requested together, one round ahead.  The counter that orders a wave's memory
  // operations is in-order and counts stores, so the first use of a freshly loaded query waits for every row the wave
  // has stored before it: with one load per batch that is a full store drain per batch -- with the tables in LDS the
  // only thing the wave ever waits for (8-lane rows: 1.5 of the 2.2 us a batch took).  Now once per QB batches.
  constexpr int QB = 4;
  uint64_t round0 = ((uint64_t)blockIdx.x * WAVES + (tid >> 6)) * 64u;
  T xq[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const uint64_t p = round0 + (uint64_t)j * wave_step + lane;
    xq[j] = A.q[p < A.nq ? p : A.nq - 1u];           // unconditional (clamped) loads: the compiler can count them
  }
  for (; round0 < limit; round0 += (uint64_t)QB * wave_step) {
  T xc[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j)                       // (a lane past the batch's end searches k0)
    xc[j] = (round0 + (uint64_t)j * wave_step + lane < limit) ? xq[j] : k0;
#pragma unroll
  for (int j = 0; j < QB; ++j) {                     // the next round, in flight during this one
    const uint64_t p = round0 + (uint64_t)(QB + j) * wave_step + lane;
    xq[j] = A.q[p < A.nq ? p : A.nq - 1u];
  }
#pragma unroll 1
  for (int jb = 0; jb < QB; ++jb) {
    const uint64_t base = round0 + (uint64_t)jb * wave_step;
    if (base >= limit) break;
    T x = xc[0];
#pragma unroll
    for (int j = 1; j < QB; ++j)
      if (jb == j) x = xc[j];
    const bool inr = (k0 <= x) && (x <= kn);
    if (A.check && base + lane < limit && lane_query_fails<T>(x, k0, kn, A.mode))   // fresh output: the range test rides along
      atomicMin(A.first_fail, (unsigned long long)(base + lane));
    T xs = x;
    if (A.mode == EX_PERIODIC && !inr) xs = rem_euclid_t(x - k0, kn - k0) + k0;   // cubic_spline.rs:805-809
    // (queries at / after the first failing one never get here; an inactive lane searches k0)
    uint32_t i;
    T xl, xr;
    if constexpr (GK) {
      if (A.bx32.lut) {
        const uint32_t ub = lut32_count_le<T>(A.pyr.lv0, n, A.bx32.lut, A.bx32.m, A.bx32.scale, k0, kn, xs);
        i = (ub == 0) ? 0u : ub - 1u;
        if (i > n - 2u) i = n - 2u;
      } else {
        i = locate_index<T, const T*>(A.pyr, k0, kn, xs, lane);   // all 64 lanes take part
      }
      xl = A.pyr.lv0[i];
      xr = A.pyr.lv0[i + 1];
    } else {
#ifdef NDI_TUNING
      if (A.debug & 1) i = (uint32_t)((base + lane) * 2654435761ull >> 7) % (n - 1u);
      else
#endif
      i = lut ? locate_index_lut<T>(P, lut, A.bx.m, A.bx.scale, k0, kn, xs)
              : locate_index<T, lds_ptr<T>>(P, k0, kn, xs, lane);   // all 64 lanes take part
      xl = P.lv0[i];
      xr = P.lv0[i + 1];
    }
    w_i[lane] = NDI_CHK(i, n - 1u, BC_INTERVAL) * RS;   // offset of the interval's operands (n * RS < 2^32: host)
    if (STRAT == ST_CUBIC) {
      w_c0[lane] = (xs - xl) / (xr - xl);   // t, cubic_spline.rs:818
      if (TLDS == 2) w_c1[lane] = xr - xl;  // dx of the interval, as the build formed it (spline_dx_up_kernel)
    } else {
      w_c0[lane] = xr - xl;                 // linear.rs:33-35: (x2 - x1), (x - x1)
      w_c1[lane] = x - xl;
    }
    __builtin_amdgcn_wave_barrier();        // LDS operations of one wave execute in order: no s_barrier needed
    const uint32_t nq_here = (limit - base < 64u) ? (uint32_t)(limit - base) : 64u;
    const uint32_t items = nq_here * LV;
    T* const o_base = A.out + base * A.out_stride;
    for (uint32_t it0 = lane; it0 < items; it0 += 64u * UNR) {
      bool live[UNR];
      uint32_t ql[UNR], v[UNR], ii[UNR];
      T s0[UNR], s1[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {        // phase 1: (query, vector) of the item and the query's scalars
        const uint32_t it = it0 + (uint32_t)k * 64u;
        live[k] = it < items;
        const uint32_t itc = live[k] ? it : 0u;
        ql[k] = (LV == 1u) ? itc : __umulhi(itc, A.lv_magic);
        ql[k] = NDI_CHK(ql[k], 64u, BC_STRIP);
        v[k] = itc - ql[k] * LV;
        ii[k] = w_i[ql[k]];
        s0[k] = w_c0[ql[k]];
        s1[k] = STRIP2 ? w_c1[ql[k]] : T(0);
      }
      V yl[UNR], yr[UNR], a[UNR], b[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {        // phase 2: the four operand vectors
#ifdef NDI_TUNING
        if (A.debug & 2) ii[k] = (lane & 7u) * RS;
        if (A.debug & 8) { yl[k] = V(s0[k]); yr[k] = V(s1[k]); a[k] = V(T(v[k])); b[k] = V(T(ql[k])); continue; }
#endif
        const uint32_t e = ii[k] + v[k];
        yl[k] = t_y[e];
        yr[k] = t_y[e + LV];
        if (TLDS == 2) {
          a[k] = t_a[e];          // k[i], k[i+1]: a / b are re-formed below
          b[k] = t_a[e + LV];
        } else if (STRAT == ST_CUBIC) {
          a[k] = t_a[e];
          b[k] = t_b[e];
        } else {
          a[k] = V(0);
          b[k] = V(0);
        }
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) {        // phase 3: polynomial and store
        RowCoef<T, STRAT> c;
        if (STRAT == ST_CUBIC) {
          const T one = T(1);
          c.c0 = one - s0[k];
          c.c1 = s0[k];
          c.c2 = s0[k] * (one - s0[k]);
        } else {
          c.c0 = s0[k];
          c.c1 = s1[k];
          c.c2 = T(0);
        }
        if (TLDS == 2) {       // cubic_spline.rs:354-365, the operations of the build's epilogue
          const V dy = yr[k] - yl[k];
          const V dxv = V(s1[k]);
          const V ak = a[k] * dxv - dy;
          const V bk = dy - b[k] * dxv;
          a[k] = ak;
          b[k] = bk;
        }
        // rows back to back (the usual case): item `it` of the batch is vector `it` behind the batch's first row
        V* o = contig ? reinterpret_cast<V*>(o_base) + (it0 + (uint32_t)k * 64u)
                      : reinterpret_cast<V*>(o_base + (uint64_t)ql[k] * A.out_stride) + v[k];
#ifdef NDI_TUNING
        if (A.debug & 4) {   // keep the arithmetic alive without the store
          const V r = row_point<T, STRAT, V>(c, yl[k], yr[k], a[k], b[k]);
          bool hit;
          if constexpr (VEC == 1) hit = r == T(-123.456); else hit = r[0] == T(-123.456);
          if (live[k] && hit) store_stream<true>(o, r);
          continue;
        }
#endif
        if (live[k]) store_stream<true>(o, row_point<T, STRAT, V>(c, yl[k], yr[k], a[k], b[k]));
      }
    }
    __builtin_amdgcn_wave_barrier();        // the strip is rewritten by the next batch
  }
  }
}

// eval_fused_kernel's work for rows of 64 bytes to a few KiB whose tables are read from L2, with the queries of a
// workgroup round ORDERED BY INTERVAL in LDS first.  In query order every output vector costs four operand vectors from
// L2 -- random rows, an L1 that holds a thirtieth of the table set -- and the counters of f64 x 32 lanes show the L2
// REQUEST rate as the first bound: 1.8e8 64-byte requests (1.2e8 of them operand reads) in a 1.0 ms kernel against
// 2.7e11 requests / s of the chip's 128 L2 channels (profiles/r05_fused_32lane_counters.txt).  A round of TB * QPT
// queries (4096) on 1023 intervals has four queries per interval: ordered by interval, neighbouring items read the SAME
// operand rows, which then come from L1 -- the table reads from L2 drop to about one per interval and round.  The order
// is formed in LDS (ranks by LDS atomics, an exclusive scan of the interval counts, {interval, local index} and the
// query's scalars placed in interval order); rows are still written at their own positions, whole rows at a time.
// Same search, same operands, same operation order as eval_fused_kernel (TLDS == 0, plain tables).
template <class T, int STRAT, int VEC, int TB, int QPT>
__global__ __launch_bounds__(TB) void eval_fused_sorted_kernel(EvalFusedArgs<T> A) {
  using V = typename VecT<T, VEC>::type;
  constexpr bool STRIP2 = STRAT == ST_LINEAR;
  constexpr uint32_t NQ = (uint32_t)TB * QPT;
  static_assert(NQ <= 4096, "the local index takes 12 bits of the key");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ uint32_t s_wave[TB / 64];
  if (A.nq == 0) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n = A.pyr.n, n1 = A.pyr.n1, nint = n - 1u;
  // LDS: [pyramid | lut | interval counters | keys | c0 | (linear) c1]
  size_t off;
  {
    T* s0 = reinterpret_cast<T*>(smem_raw);
    const uint32_t total = n + n1;
    for (uint32_t i = tid; i < total; i += TB) s0[i] = A.pyr.lv0[i];
    off = ((size_t)total * sizeof(T) + 15u) & ~(size_t)15u;
  }
  lds_u16 lut = nullptr;
  if (A.bx.lut) {
    uint32_t* sl = reinterpret_cast<uint32_t*>(smem_raw + off);
    const uint32_t words = (A.bx.m + 2u) / 2u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(A.bx.lut);
    for (uint32_t i = tid; i < words; i += TB) sl[i] = src[i];
    lut = (lds_u16)(smem_raw + off);
    off += ((size_t)words * 4u + 15u) & ~(size_t)15u;
  }
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem_raw + off);
  off += ((size_t)nint * 4u + 15u) & ~(size_t)15u;
  uint32_t* s_key = reinterpret_cast<uint32_t*>(smem_raw + off);
  off += (size_t)NQ * 4u;
  T* s_c0 = reinterpret_cast<T*>(smem_raw + off);
  off += (size_t)NQ * sizeof(T);
  T* s_c1 = reinterpret_cast<T*>(smem_raw + off);
  for (uint32_t b = tid; b < nint; b += TB) s_cnt[b] = 0u;
  __syncthreads();
  PyramidLds<T> P;
  P.lv0 = (lds_ptr<T>)(smem_raw);
  P.lv1 = P.lv0 + n;
  P.n = n; P.n1 = n1; P.levels = A.pyr.levels; P.guess = A.pyr.guess; P.block = A.pyr.block;
  const T k0 = P.lv0[0], kn = P.lv0[n - 1];
  unsigned long long limit = A.check ? NO_FAIL : *A.first_fail;
  if (limit > A.nq) limit = A.nq;
  const uint32_t LV = A.lv;
  const V* const t_y = reinterpret_cast<const V*>(A.data);
  const V* const t_a = reinterpret_cast<const V*>(A.ca);
  const V* const t_b = reinterpret_cast<const V*>(A.cb);
  for (uint64_t base = (uint64_t)blockIdx.x * NQ; base < limit; base += (uint64_t)gridDim.x * NQ) {   // (workgroup-uniform)
    uint32_t iv[QPT], rank[QPT];
    T c0v[QPT], c1v[QPT];
    T xq[QPT];
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
      const uint64_t qi = base + (uint64_t)k * TB + tid;
      xq[k] = A.q[qi < A.nq ? qi : A.nq - 1u];
    }
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
      const uint64_t qi = base + (uint64_t)k * TB + tid;
      const bool act = qi < limit;
      const T x = act ? xq[k] : k0;
      const bool inr = (k0 <= x) && (x <= kn);
      if (A.check && act && lane_query_fails<T>(x, k0, kn, A.mode)) atomicMin(A.first_fail, (unsigned long long)qi);
      T xs = x;
      if (A.mode == EX_PERIODIC && !inr) xs = rem_euclid_t(x - k0, kn - k0) + k0;   // cubic_spline.rs:805-809
      const uint32_t i = lut ? locate_index_lut<T>(P, lut, A.bx.m, A.bx.scale, k0, kn, xs)
                             : locate_index<T, lds_ptr<T>>(P, k0, kn, xs, lane);   // all 64 lanes take part
      const T xl = P.lv0[i], xr = P.lv0[i + 1];
      iv[k] = NDI_CHK(i, nint, BC_INTERVAL);
      if (STRAT == ST_CUBIC) {
        c0v[k] = (xs - xl) / (xr - xl);   // t, cubic_spline.rs:818
        c1v[k] = T(0);
      } else {
        c0v[k] = xr - xl;                 // linear.rs:33-35: (x2 - x1), (x - x1)
        c1v[k] = x - xl;
      }
      rank[k] = act ? atomicAdd(&s_cnt[iv[k]], 1u) : 0u;
    }
    __syncthreads();
    {   // exclusive scan of the interval counts, TB intervals at a time
      uint32_t carry = 0;
      for (uint32_t b0 = 0; b0 < nint; b0 += TB) {
        const uint32_t b = b0 + tid;
        const uint32_t c = b < nint ? s_cnt[b] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
          const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
          if (lane >= d) incl += up;
        }
        __syncthreads();
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)TB / 64u; ++w) {
          const uint32_t t = s_wave[w];
          if (w < wave) wave_off += t;
          total += t;
        }
        if (b < nint) s_cnt[b] = carry + wave_off + incl - c;
        carry += total;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
      const uint64_t qi = base + (uint64_t)k * TB + tid;
      if (qi < limit) {
        const uint32_t p = NDI_CHK(s_cnt[iv[k]] + rank[k], NQ, BC_POSITION);
        s_key[p] = (iv[k] << 12) | ((uint32_t)k * TB + tid);
        s_c0[p] = c0v[k];
        if (STRIP2) s_c1[p] = c1v[k];
      }
    }
    __syncthreads();
    const uint32_t nloc = (limit - base < (uint64_t)NQ) ? (uint32_t)(limit - base) : NQ;
    const uint32_t items = nloc * LV;
    for (uint32_t it = tid; it < items; it += TB) {
      const uint32_t p = (LV == 1u) ? it : __umulhi(it, A.lv_magic);
      const uint32_t v = it - p * LV;
      const uint32_t key = s_key[NDI_CHK(p, NQ, BC_POSITION)];
      const uint32_t i = key >> 12, loc = key & 4095u;
      const T s0 = s_c0[p];
      const T s1 = STRIP2 ? s_c1[p] : T(0);
      const uint32_t e = i * LV + v;
      const V yl = t_y[e], yr = t_y[e + LV];
      V a = V(0), b = V(0);
      if (STRAT == ST_CUBIC) { a = t_a[e]; b = t_b[e]; }
      RowCoef<T, STRAT> c;
      if (STRAT == ST_CUBIC) {
        const T one = T(1);
        c.c0 = one - s0;
        c.c1 = s0;
        c.c2 = s0 * (one - s0);
      } else {
        c.c0 = s0;
        c.c1 = s1;
        c.c2 = T(0);
      }
      V* o = reinterpret_cast<V*>(A.out + (base + loc) * A.out_stride) + v;
      store_stream<true>(o, row_point<T, STRAT, V>(c, yl, yr, a, b));
    }
    __syncthreads();                                   // the round's arrays are rewritten by the next one
    for (uint32_t b = tid; b < nint; b += TB) s_cnt[b] = 0u;
    __syncthreads();
  }
}

// QUERY PER LANE with the whole table set resident in LDS -- the formulation for the reference's own bench shapes at
// large Q (scalar data and (100, 5) on 100 knots: benches/bench_interp1d.rs:12-47, 82-122).  Counters of the query-order
// kernel on those shapes (profiles/r05_small_shapes_counters.txt): 126 VALU wave instructions per 64 scalar queries, 43 %
// of the LDS cycles bank conflicts, nothing saturated -- the per-query strip round trip, one item per lane and trip,
// 8-byte loads and stores, a search with divergent branches and an IEEE division per query are overhead when a row is one
// or a few values.  Here a lane owns its query from the search to the result, and everything per query is branch-free:
//  * search: a DENSE bucket index (DenseLut: so many uniform buckets -- 4n .. 32n -- that no bucket holds more than
//    `maxk` <= 8 knots, built on the host with the device's arithmetic): count = lut[bucket(x)] + sum over the next
//    maxk knots of (k <= x) -- knots past the bucket are > x because bucket() is monotone -- i.e. one 16-bit read and
//    maxk knot reads, no loop with a data-dependent trip count, no cross-lane step.  Axes whose O(1) formula guess is
//    right for every x (the default index axis, linspace) need no index at all (vector_extensions.rs:68-90);
//  * staging (once per workgroup): knots, index, one record {x_l, dx, RN(1 / dx)} per interval (cubic) and one record
//    per (interval, lane of the trailing axes): {y_l, y_r, a, b} (cubic) or {y_l, m} (linear) with
//    m = (y_r - y_l) / (x_r - x_l) -- Linear::calc_frac's division (linear.rs:33) has no query in it, so it is done
//    once per record with the IEEE division: the same operands, the same bits;
//  * t = (x - x_l) / dx (cubic_spline.rs:818) by the correctly rounded shared-divisor division (div_shared: the bits of
//    the IEEE division; the reciprocal comes from the interval's record);
//  * eval_scalar_kernel (L == 1): QPL consecutive queries per lane, one 16-byte query load and one 16-byte store per
//    lane, the next vector of queries in flight;
//  * eval_lanes_kernel (2 <= L, rows of up to 64 bytes): the wave's 64 rows are written to a wave-private LDS strip and
//    leave as ONE sequential stream of 16-byte vectors (64 * L * sizeof(T) consecutive bytes per batch), whatever L is.
// Same operations in the same order as Linear / CubicSplineStrategy::interp_into; rows at / after the batch's first
// failing query (range_check_kernel) are never written.
// Records in LDS, NS values each, kept as a STRUCTURE OF ARRAYS of 16-byte units (NS * sizeof(T) / 16 arrays, 16-byte
// stride): a gather of one unit by 64 lanes then spreads over all LDS banks.  Records kept whole -- four doubles, 32 bytes
// apart -- put every lane on one of four bank groups: the first version of these kernels spent 2.0e8 LDS conflict cycles
// per 1e8 scalar f64 queries, 0.33 of its 0.67 ms (profiles/r05_lanes_counters.txt); f32 records were 16 bytes already.
template <class T, int NS>
struct RecArr {
  static constexpr int EL = 16 / (int)sizeof(T);
  static constexpr int NU = NS / EL;
  static_assert(NS % EL == 0, "whole 16-byte units");
  using U = typename VecT<T, EL>::type;
  U* base;
  uint32_t count;
  __device__ __forceinline__ void put(uint32_t i, const T (&v)[NS]) const {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      U w;
#pragma unroll
      for (int e = 0; e < EL; ++e) w[e] = v[u * EL + e];
      base[(size_t)u * count + i] = w;
    }
  }
  __device__ __forceinline__ void get(uint32_t i, T (&v)[NS]) const {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const U w = base[(size_t)u * count + i];
#pragma unroll
      for (int e = 0; e < EL; ++e) v[u * EL + e] = w[e];
    }
  }
  __host__ __device__ static constexpr size_t bytes(size_t count) { return (size_t)NU * count * 16u; }
};
template <class T>
using XRecs = RecArr<T, 4>;            // {x_l, dx, r = RN(1 / dx) or 0 outside the shared-divisor window, 0} per interval
template <class T, int STRAT>
struct TabRecs {                       // cubic {y_l, y_r, a, b}; linear {y_l, m} (padded to a whole unit)
  static constexpr int NS = STRAT == ST_CUBIC ? 4 : (16 / (int)sizeof(T) > 2 ? 16 / (int)sizeof(T) : 2);
  using type = RecArr<T, NS>;
};

template <class T>
struct DenseLut {
  const uint16_t* lut;   // m + 1 entries (global memory); nullptr: the axis' O(1) formula guess is exact
  uint32_t m, maxk;
  T scale;               // m / (kn - k0), in T
};

template <class T>
struct EvalLanesArgs {
  const T* knots;        // [n]
  uint32_t n;
  DenseLut<T> dl;
  const T* data;         // [n][lanes]
  const T* ca;           // [n-1][lanes] (cubic)
  const T* cb;
  const T* q;
  T* out;
  uint64_t nq, out_stride;
  uint32_t lanes;
  int mode;              // ExtrapMode
  unsigned long long* first_fail;
  int check;             // 1: no range pre-pass ran (NDI_EVAL_FRESH_OUTPUT): the kernel tests every query itself, records
                         //    the lowest failing index and evaluates all rows; 0: *first_fail is final, rows at / after it
                         //    are not written
};

// The failure condition of a query (Interp1D::is_in_range, interp1d/mod.rs:384-386; NaN with extrapolation: the
// reference panics, vector_extensions.rs:83-84) -- range_check_kernel's test, for kernels that check on the fly.
template <class T>
__device__ __forceinline__ bool lane_query_fails(T x, T k0, T kn, int mode) {
  const bool inr = (k0 <= x) && (x <= kn);
  if (mode == EX_NO) return !inr;
  if (mode == EX_PERIODIC && !inr) {           // +-inf wraps to NaN, as in locate_slice
    const T xs = rem_euclid_t(x - k0, kn - k0) + k0;
    return !(xs == xs);
  }
  return !(x == x);
}

// LDS-resident state of the branch-free search
template <class T>
struct LaneAxis {
  lds_ptr<T> k;
  lds_u16 lut;           // nullptr: O(1) guess
  uint32_t n, m, maxk;
  T k0, kn, scale, gfac; // gfac = (n - 1) / (kn - k0): the guess's factor (vector_extensions.rs:70-90)
};

constexpr uint32_t LANE_SENTINELS = 8;   // +inf entries staged behind the knots: k[lo + j] needs no bound for j < maxk <= 8

// how many of k[lo .. lo + MK) are <= x: MK independent LDS reads issued together, then MK compares
template <class T, int MK>
__device__ __forceinline__ uint32_t knots_le(lds_ptr<T> k, uint32_t lo, T x) {
  T v[MK];
#pragma unroll
  for (int j = 0; j < MK; ++j) v[j] = k[lo + j];
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < MK; ++j) c += (v[j] <= x) ? 1u : 0u;
  return c;
}

// the unique i with k[i] <= x < k[i+1], clamped to [0, n-2]; NaN -> 0
template <class T>
__device__ __forceinline__ uint32_t lane_axis_index(const LaneAxis<T>& S, T x) {
  if (S.lut) {                                            // (workgroup-uniform)
    T f = (x - S.k0) * S.scale;
    f = fmax(f, T(0));                                    // below the axis, NaN -> bucket 0
    f = fmin(f, T(S.m - 1u));
    const uint32_t lo = S.lut[(uint32_t)f];
    uint32_t cnt = lo;
    switch (S.maxk) {                                     // (uniform) knots of later buckets -- and the sentinels -- are > x
      case 1: cnt += knots_le<T, 1>(S.k, lo, x); break;
      case 2: cnt += knots_le<T, 2>(S.k, lo, x); break;
      case 3: cnt += knots_le<T, 3>(S.k, lo, x); break;
      case 4: cnt += knots_le<T, 4>(S.k, lo, x); break;
      default: cnt += knots_le<T, 8>(S.k, lo, x); break;
    }
    const uint32_t i = cnt ? cnt - 1u : 0u;               // (x = +inf counts sentinels too: the clamp below covers it)
    return i < S.n - 2u ? i : S.n - 2u;
  }
  const T mm = S.gfac * (x - S.k0) + T(0);                // locate_index's guess, verified exact for every x on the host
  return (mm >= T(0)) ? (uint32_t)(mm < T(S.n - 2u) ? mm : T(S.n - 2u)) : 0u;
}

// stages knots + dense index and returns the axis view; `off` advances past them (16-byte aligned)
template <class T, int TB>
__device__ __forceinline__ LaneAxis<T> stage_lane_axis(unsigned char* smem, size_t& off, const T* knots, uint32_t n,
                                                       const DenseLut<T>& dl) {
  const uint32_t tid = threadIdx.x;
  T* s0 = reinterpret_cast<T*>(smem + off);
  for (uint32_t i = tid; i < n; i += TB) s0[i] = knots[i];
  if (tid < LANE_SENTINELS) s0[n + tid] = __builtin_huge_val();   // +inf (converted to T)
  LaneAxis<T> S;
  S.k = (lds_ptr<T>)(smem + off);
  off += ((size_t)(n + LANE_SENTINELS) * sizeof(T) + 15u) & ~(size_t)15u;
  S.lut = nullptr;
  if (dl.lut) {
    uint32_t* sl = reinterpret_cast<uint32_t*>(smem + off);
    const uint32_t words = (dl.m + 2u) / 2u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(dl.lut);
    for (uint32_t i = tid; i < words; i += TB) sl[i] = src[i];
    S.lut = (lds_u16)(smem + off);
    off += ((size_t)words * 4u + 15u) & ~(size_t)15u;
  }
  S.n = n; S.m = dl.m; S.maxk = dl.maxk; S.scale = dl.scale;
  S.k0 = knots[0];
  S.kn = knots[n - 1];
  S.gfac = (T(n - 1u) - T(0)) / (S.kn - S.k0);
  return S;
}

// interval records {x_l, dx, RN(1 / dx)} of an axis
template <class T, int TB>
__device__ __forceinline__ void stage_xrecs(const XRecs<T>& X, const T* knots, uint32_t n) {
  for (uint32_t i = threadIdx.x; i + 1u < n; i += TB) {
    const T xl = knots[i], xr = knots[i + 1];
    const SharedDivisor<T> sd = shared_divisor<T>(xr - xl);
    const T v[4] = {xl, sd.d, sd.ok ? sd.r : T(0), T(0)};
    X.put(i, v);
  }
}

// table records: cubic {y_l, y_r, a, b}, linear {y_l, m}
template <class T, int STRAT, int TB>
__device__ __forceinline__ void stage_table_recs(const typename TabRecs<T, STRAT>::type& R, const EvalLanesArgs<T>& A) {
  constexpr int NS = TabRecs<T, STRAT>::NS;
  const uint32_t L = A.lanes, total = (A.n - 1u) * L;
  for (uint32_t e = threadIdx.x; e < total; e += TB) {
    const T yl = A.data[e], yr = A.data[e + L];
    T v[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = T(0);
    v[0] = yl;
    if (STRAT == ST_CUBIC) {
      v[1] = yr;
      v[2] = A.ca[e];
      v[3] = A.cb[e];
    } else {
      const uint32_t i = e / L;
      const T dx = A.knots[i + 1] - A.knots[i];
      v[1] = (yr - yl) / dx;                              // linear.rs:33, once per record
    }
    R.put(e, v);
  }
}

// One query: interval, then the per-query scalar -- t (cubic_spline.rs:818) or (x - x1) (linear.rs:35).
template <class T, int STRAT>
__device__ __forceinline__ void lane_query(const LaneAxis<T>& S, const XRecs<T>& X, int mode, T x, uint32_t& i, T& s0) {
  T xs = x;
  if (STRAT == ST_CUBIC && mode == EX_PERIODIC) {         // (uniform) cubic_spline.rs:805-809
    const bool inr = (S.k0 <= x) && (x <= S.kn);
    if (!inr) xs = rem_euclid_t(x - S.k0, S.kn - S.k0) + S.k0;
  }
  i = lane_axis_index<T>(S, xs);
  i = NDI_CHK(i, S.n - 1u, BC_INTERVAL);
  if (STRAT == ST_CUBIC) {
    T xr[4];
    X.get(i, xr);
    SharedDivisor<T> sd;
    sd.d = xr[1]; sd.r = xr[2]; sd.ok = xr[2] > T(0);
    s0 = div_shared<T, T>(xs - xr[0], sd);
  } else {
    s0 = x - S.k[i];
  }
}

template <class T, int STRAT>
__device__ __forceinline__ T lane_point(const typename TabRecs<T, STRAT>::type& R, uint32_t rec, T s0) {
  T r[TabRecs<T, STRAT>::NS];
  R.get(rec, r);
  if (STRAT == ST_CUBIC) {                                // cubic_spline.rs:825-827
    const T yl = r[0], yr = r[1], a = r[2], b = r[3];
    const T c0 = T(1) - s0;
    return c0 * yl + s0 * yr + (s0 * c0) * (a * c0 + b * s0);
  } else {                                                // linear.rs:33-35 with the record's m
    return r[1] * s0 + r[0];
  }
}

// NQ independent scalar queries of one lane in LOCKSTEP: every stage is done for all of them before the next begins -- NQ
// bucket reads, then NQ x MK knot reads, then the NQ interval records, the NQ table records, the NQ polynomials -- so a
// wave has NQ LDS reads in flight per dependent step instead of one (one query after the other left the SIMDs 21 % busy and
// the memory pipe at 3 TB/s: every query is a chain of five dependent LDS round trips; profiles/r05_lanes_counters.txt).
// The only data-dependent branch -- a quotient outside the shared-divisor window redone with the IEEE division -- sits
// behind everything else.  MK = the index's knots per bucket (0: the axis' exact O(1) guess).
template <class T, int STRAT, int NQ, int MK>
__device__ __forceinline__ void eval_scalar_queries(const LaneAxis<T>& S, const XRecs<T>& X,
                                                    const typename TabRecs<T, STRAT>::type& R, int mode, const T (&x)[NQ],
                                                    T (&out)[NQ]) {
  T xs[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) xs[q] = x[q];
  if (STRAT == ST_CUBIC && mode == EX_PERIODIC) {         // (uniform) cubic_spline.rs:805-809
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool inr = (S.k0 <= x[q]) && (x[q] <= S.kn);
      if (!inr) xs[q] = rem_euclid_t(x[q] - S.k0, S.kn - S.k0) + S.k0;
    }
  }
  uint32_t iv[NQ];
  if constexpr (MK > 0) {
    uint32_t lo[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      T f = (xs[q] - S.k0) * S.scale;
      f = fmax(f, T(0));
      f = fmin(f, T(S.m - 1u));
      lo[q] = S.lut[(uint32_t)f];
    }
    T kv[NQ][MK];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < MK; ++j) kv[q][j] = S.k[lo[q] + j];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      uint32_t cnt = lo[q];
#pragma unroll
      for (int j = 0; j < MK; ++j) cnt += (kv[q][j] <= xs[q]) ? 1u : 0u;
      const uint32_t i = cnt ? cnt - 1u : 0u;
      iv[q] = NDI_CHK(i < S.n - 2u ? i : S.n - 2u, S.n - 1u, BC_INTERVAL);
    }
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const T mm = S.gfac * (xs[q] - S.k0) + T(0);
      iv[q] = (mm >= T(0)) ? (uint32_t)(mm < T(S.n - 2u) ? mm : T(S.n - 2u)) : 0u;
    }
  }
  T s0[NQ];
  bool redo[NQ];
  SharedDivisor<T> sd[NQ];
  T num[NQ];
  if (STRAT == ST_CUBIC) {
    T xr[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) X.get(iv[q], xr[q]);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      sd[q].d = xr[q][1]; sd[q].r = xr[q][2]; sd[q].ok = xr[q][2] > T(0);
      num[q] = xs[q] - xr[q][0];
      bool ok;
      s0[q] = div_shared_fast<T>(num[q], sd[q], ok);      // t, cubic_spline.rs:818
      redo[q] = !ok;
    }
  } else {
    T xl[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) xl[q] = S.k[iv[q]];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      s0[q] = x[q] - xl[q];                               // linear.rs:35's (x - x1)
      redo[q] = false;
    }
  }
  T r[NQ][TabRecs<T, STRAT>::NS];
#pragma unroll
  for (int q = 0; q < NQ; ++q) R.get(iv[q], r[q]);
  if (STRAT == ST_CUBIC) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (__builtin_expect(redo[q], 0)) s0[q] = num[q] / sd[q].d;   // outside the window: the IEEE division
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (STRAT == ST_CUBIC) {                              // cubic_spline.rs:825-827
      const T yl = r[q][0], yr = r[q][1], a = r[q][2], b = r[q][3];
      const T c0 = T(1) - s0[q];
      out[q] = c0 * yl + s0[q] * yr + (s0[q] * c0) * (a * c0 + b * s0[q]);
    } else {                                              // linear.rs:33-35 with the record's m
      out[q] = r[q][1] * s0[q] + r[q][0];
    }
  }
}

template <class T, int STRAT, int QPL, int TB>
__global__ __launch_bounds__(TB) void eval_scalar_kernel(EvalLanesArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using QV = typename VecT<T, QPL>::type;
  using Tab = typename TabRecs<T, STRAT>::type;
  if (A.nq == 0) return;
  const uint32_t tid = threadIdx.x, n = A.n;
  size_t off = 0;
  const LaneAxis<T> S = stage_lane_axis<T, TB>(smem_raw, off, A.knots, n, A.dl);
  XRecs<T> X{reinterpret_cast<typename XRecs<T>::U*>(smem_raw + off), n - 1u};
  if (STRAT == ST_CUBIC) {
    stage_xrecs<T, TB>(X, A.knots, n);
    off += XRecs<T>::bytes(n - 1u);
  }
  const Tab R{reinterpret_cast<typename Tab::U*>(smem_raw + off), n - 1u};
  stage_table_recs<T, STRAT, TB>(R, A);
  __syncthreads();
  unsigned long long limit = A.check ? NO_FAIL : *A.first_fail;
  if (limit > A.nq) limit = A.nq;
  // QPL consecutive queries per lane, vector load / store; NV vectors (NQ = 4 queries) per thread and trip, evaluated in
  // lockstep (eval_scalar_queries), the next trip's vectors in flight; the (< QPL) queries behind the last full vector
  // below `limit` go one per thread at the end
  constexpr int NQ = QPL >= 4 ? QPL : 4;
  constexpr int NV = NQ / QPL;
  const uint64_t nvec = limit / QPL;
  const uint64_t step = (uint64_t)gridDim.x * TB;
  const QV* qv = reinterpret_cast<const QV*>(A.q);
  QV* ov = reinterpret_cast<QV*>(A.out);
  const uint64_t vlast = nvec ? nvec - 1u : 0u;
  auto loadq = [&](uint64_t v) -> QV { return nvec ? qv[v < nvec ? v : vlast] : QV(S.k0); };   // clamped: unconditional
  auto run = [&](const T (&x)[NQ], T (&res)[NQ]) {        // (uniform dispatch on the index's knots per bucket)
    if (!S.lut) eval_scalar_queries<T, STRAT, NQ, 0>(S, X, R, A.mode, x, res);
    else if (S.maxk == 1) eval_scalar_queries<T, STRAT, NQ, 1>(S, X, R, A.mode, x, res);
    else if (S.maxk == 2) eval_scalar_queries<T, STRAT, NQ, 2>(S, X, R, A.mode, x, res);
    else if (S.maxk == 3) eval_scalar_queries<T, STRAT, NQ, 3>(S, X, R, A.mode, x, res);
    else if (S.maxk == 4) eval_scalar_queries<T, STRAT, NQ, 4>(S, X, R, A.mode, x, res);
    else eval_scalar_queries<T, STRAT, NQ, 8>(S, X, R, A.mode, x, res);
  };
  uint64_t vi = (uint64_t)blockIdx.x * TB + tid;
  QV nxt[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) nxt[k] = loadq(vi + (uint64_t)k * step);
  for (; vi < nvec; vi += (uint64_t)NV * step) {
    T x[NQ], res[NQ];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int u = 0; u < QPL; ++u) {
        if constexpr (QPL == 1) x[k] = nxt[k]; else x[k * QPL + u] = nxt[k][u];
      }
#pragma unroll
    for (int k = 0; k < NV; ++k) nxt[k] = loadq(vi + (uint64_t)(NV + k) * step);
    run(x, res);
    if (A.check) {                                        // (uniform) fresh output: the range test rides along
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
          const uint64_t v = vi + (uint64_t)k * step;
          if (v < nvec && lane_query_fails<T>(x[k * QPL + u], S.k0, S.kn, A.mode))
            atomicMin(A.first_fail, (unsigned long long)(v * QPL + u));
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const uint64_t v = vi + (uint64_t)k * step;
      if (v < nvec) {
        if constexpr (QPL == 1) {
          store_stream<true>(A.out + v * A.out_stride, res[k]);   // (rows of one value may be strided)
        } else {
          QV w;
#pragma unroll
          for (int u = 0; u < QPL; ++u) w[u] = res[k * QPL + u];
          store_stream<true>(ov + v, w);
        }
      }
    }
  }
  if constexpr (QPL > 1) {
    const uint64_t qi = nvec * QPL + tid;
    if (blockIdx.x == 0 && qi < limit) {
      uint32_t i;
      T s0;
      const T xq = A.q[qi];
      if (A.check && lane_query_fails<T>(xq, S.k0, S.kn, A.mode)) atomicMin(A.first_fail, (unsigned long long)qi);
      lane_query<T, STRAT>(S, X, A.mode, xq, i, s0);
      A.out[qi * A.out_stride] = lane_point<T, STRAT>(R, i, s0);
    }
  }
}

// LC: the trailing axis' length when it is one of the instantiated values (the per-lane loop unrolls), else 0
template <class T, int STRAT, int LC, int TB>
__global__ __launch_bounds__(TB) void eval_lanes_kernel(EvalLanesArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int VN = Wide<T>::N;
  using V = typename VecT<T, VN>::type;
  using Tab = typename TabRecs<T, STRAT>::type;
  if (A.nq == 0) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, n = A.n;
  const uint32_t L = LC ? (uint32_t)LC : A.lanes;
  size_t off = 0;
  const LaneAxis<T> S = stage_lane_axis<T, TB>(smem_raw, off, A.knots, n, A.dl);
  XRecs<T> X{reinterpret_cast<typename XRecs<T>::U*>(smem_raw + off), n - 1u};
  if (STRAT == ST_CUBIC) {
    stage_xrecs<T, TB>(X, A.knots, n);
    off += XRecs<T>::bytes(n - 1u);
  }
  const Tab R{reinterpret_cast<typename Tab::U*>(smem_raw + off), (n - 1u) * L};
  stage_table_recs<T, STRAT, TB>(R, A);
  off += Tab::bytes((size_t)(n - 1u) * L);
  T* s_strip = reinterpret_cast<T*>(smem_raw + off) + (size_t)(tid >> 6) * 64u * L;
  __syncthreads();
  unsigned long long limit = A.check ? NO_FAIL : *A.first_fail;
  if (limit > A.nq) limit = A.nq;
  // (16-byte vector stores of the batch's rows: the rows must be contiguous AND the buffer 16-byte aligned -- a sliced
  //  view such as out[1:] with 5 lanes takes the per-element path)
  const bool contig = A.out_stride == (uint64_t)L && (reinterpret_cast<uintptr_t>(A.out) & 15u) == 0u;
  // even L: lane j writes its row starting at value (j mod L) -- a plain l = 0, 1, ... order would put the lanes of a
  // wave on L-element strides, i.e. (for L = 8 doubles) on two bank groups; odd strides spread by themselves
  const uint32_t rot = (L & 1u) ? 0u : lane % L;
  const uint64_t wstep = (uint64_t)gridDim.x * TB;
  uint64_t base = ((uint64_t)blockIdx.x * (TB / 64) + (tid >> 6)) * 64u;
  T xn = A.q[(base + lane < A.nq) ? base + lane : A.nq - 1u];
  for (; base < limit; base += wstep) {
    const T x = xn;                                       // (a lane past `limit` computes a row nobody stores)
    {
      const uint64_t pn = base + wstep + lane;
      xn = A.q[pn < A.nq ? pn : A.nq - 1u];               // the next batch, in flight during this one
    }
    if (A.check && base + lane < limit && lane_query_fails<T>(x, S.k0, S.kn, A.mode))
      atomicMin(A.first_fail, (unsigned long long)(base + lane));
    uint32_t i;
    T s0;
    lane_query<T, STRAT>(S, X, A.mode, x, i, s0);
    const uint32_t rec0 = i * L;
    T* mine = s_strip + lane * L;
#pragma unroll
    for (uint32_t k = 0; k < L; ++k) {
      uint32_t l = k + rot;
      if (l >= L) l -= L;
      mine[l] = lane_point<T, STRAT>(R, rec0 + l, s0);
    }
    __builtin_amdgcn_wave_barrier();                      // LDS operations of one wave execute in order
    const uint32_t nq_here = (limit - base < 64u) ? (uint32_t)(limit - base) : 64u;
    const uint32_t total = nq_here * L;
    if (contig) {
      T* const o = A.out + base * L;                      // 64 * L * sizeof(T) bytes per batch: 16-byte aligned with `out`
      for (uint32_t e0 = lane * VN; e0 < total; e0 += 64u * VN) {
        if (e0 + VN <= total) {
          store_stream<true>(reinterpret_cast<V*>(o + e0), *reinterpret_cast<const V*>(s_strip + e0));
        } else {
          for (uint32_t e = e0; e < total; ++e) o[e] = s_strip[e];
        }
      }
    } else {
      for (uint32_t it = lane; it < total; it += 64u) {
        const uint32_t ql = it / L, l = it - ql * L;
        A.out[(base + ql) * A.out_stride + l] = s_strip[it];
      }
    }
    __builtin_amdgcn_wave_barrier();                      // the strip is rewritten by the next batch
  }
}

// ---------------------------------------------------------------------------------------------
// BUCKETED: counting sort of the valid queries by interval, then a streaming evaluation
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void bucket_count_kernel(const uint32_t* idx, uint64_t nq,
                                                             const StatusBlock* status,
                                                             uint32_t* counts) {
  const uint64_t limit = nq;  // every query is grouped; the evaluation skips qi >= first_fail
  for (uint64_t qi = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; qi < limit;
       qi += (uint64_t)gridDim.x * BLOCK)
    atomicAdd(&counts[idx[qi]], 1u);
}

// Single-workgroup exclusive scan of nb counters -> cursor[] (start offsets); n_valid = total.
// TB = 256 when the kernel has to find room beside a running evaluation kernel (ring pipeline): a 4-wave workgroup
// fits wherever one evaluation workgroup has just retired, a 16-wave one waits for a quarter of a CU to drain.
template <int TB>
__global__ __launch_bounds__(TB) void bucket_scan_kernel(const uint32_t* counts, uint32_t nb,
                                                         uint32_t* cursor, StatusBlock* status) {
  __shared__ uint32_t part[TB];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (nb + (uint32_t)TB - 1u) / (uint32_t)TB;
  const uint32_t b0 = tid * per;
  uint32_t s = 0;
  for (uint32_t k = 0; k < per; ++k)
    if (b0 + k < nb) s += counts[b0 + k];
  part[tid] = s;
  __syncthreads();
  for (uint32_t off = 1; off < (uint32_t)TB; off <<= 1) {  // Hillis-Steele inclusive scan
    uint32_t v = (tid >= off) ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = (tid == 0) ? 0u : part[tid - 1];
  for (uint32_t k = 0; k < per; ++k)
    if (b0 + k < nb) {
      cursor[b0 + k] = run;
      run += counts[b0 + k];
    }
  if (tid == (uint32_t)TB - 1u) status->n_valid = part[TB - 1];
}

template <class T>
__global__ __launch_bounds__(BLOCK) void bucket_scatter_kernel(const uint32_t* idx, const T* sval, uint64_t nq,
                                                               const StatusBlock* status,
                                                               uint32_t* cursor, uint4* rec) {
  const uint64_t limit = nq;
  for (uint64_t qi = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; qi < limit;
       qi += (uint64_t)gridDim.x * BLOCK) {
    const uint32_t i = idx[qi];
    const uint32_t pos = atomicAdd(&cursor[i], 1u);
    rec[pos] = make_rec<T>((uint32_t)qi, i, sval[qi]);
  }
}

// Block-local counting sort (used when the interval histogram fits LDS): locate_kernel leaves one
// histogram per query slice in hist[B][nb]; group_offsets_kernel turns column b-prefixes into per-slice
// start offsets (one thread per interval, coalesced over intervals) and the interval totals;
// bucket_scan_kernel scans the totals; group_scatter_kernel re-reads each slice and places its queries
// with LDS atomics only.  No global atomics, deterministic output order within a bucket per slice.
__global__ __launch_bounds__(BLOCK) void group_offsets_kernel(uint32_t* hist, uint32_t nblocks, uint32_t nb,
                                                              uint32_t* totals) {
  const uint32_t bin = blockIdx.x * BLOCK + threadIdx.x;
  if (bin >= nb) return;
  uint32_t run = 0;
  uint32_t b = 0;
  for (; b + 8 <= nblocks; b += 8) {
    uint32_t h[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) h[u] = hist[(uint64_t)(b + u) * nb + bin];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      hist[(uint64_t)(b + u) * nb + bin] = run;
      run += h[u];
    }
  }
  for (; b < nblocks; ++b) {
    const uint32_t h = hist[(uint64_t)b * nb + bin];
    hist[(uint64_t)b * nb + bin] = run;
    run += h;
  }
  totals[bin] = run;
}

// Exclusive scan of the nb bin totals into cursor[] (= bin_start[]; cursor2: a second, consumable copy) by ONE workgroup
// of TB threads with coalesced 16-byte pieces -- thread-local prefix of 4, wave scan by cross-lane shifts, the wave totals
// through LDS -- and, while every bin's [start, end) is in registers, for every chunk of `chunk` grouped positions the
// tile that holds the chunk's first position (what tile_chunk_bins_kernel found by bisection).  totals / cursor must be
// allocated with nb rounded up to a multiple of 4 entries.
template <int TB>
__device__ __forceinline__ void scan_bin_totals(const uint32_t* totals, uint32_t nb, uint32_t* cursor, uint32_t* cursor2,
                                                StatusBlock* status, uint32_t chunk, uint32_t* chunk_bin,
                                                uint32_t* s_wave) {
  static_assert(TB >= 64 && TB <= 1024 && 4096 % TB == 0, "one to sixteen waves");
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  uint32_t carry = 0;
  constexpr int NT = 4096 / TB;    // tiles of TB * 4 bins whose loads are issued together (16 384 bins: all at once)
  const int chunk_shift = (chunk && (chunk & (chunk - 1u)) == 0u) ? (31 - __builtin_clz(chunk)) : -1;   // power of two: shifts
  for (uint32_t base = 0; base < nb; base += (uint32_t)TB * 4u * NT) {
    uint4 v[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const uint32_t i0 = base + ((uint32_t)j * TB + tid) * 4u;
      v[j] = i0 < nb ? reinterpret_cast<const uint4*>(totals)[i0 >> 2] : make_uint4(0u, 0u, 0u, 0u);
      if (i0 + 1u >= nb) v[j].y = 0u;     // (the padding entries of the allocation hold nothing)
      if (i0 + 2u >= nb) v[j].z = 0u;
      if (i0 + 3u >= nb) v[j].w = 0u;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const uint32_t i0 = base + ((uint32_t)j * TB + tid) * 4u;
      const uint32_t local = v[j].x + v[j].y + v[j].z + v[j].w;
      uint32_t incl = local;
#pragma unroll
      for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += up;
      }
      __syncthreads();              // (s_wave of the previous tile has been read)
      if (lane == 63u) s_wave[wave] = incl;
      __syncthreads();
      uint32_t wave_off = 0, tile_total = 0;
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)TB / 64u; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wave) wave_off += t;
        tile_total += t;
      }
      const uint32_t e0 = carry + wave_off + incl - local;
      const uint32_t e1 = e0 + v[j].x, e2 = e1 + v[j].y, e3 = e2 + v[j].z, e4 = e3 + v[j].w;
      if (i0 < nb) reinterpret_cast<uint4*>(cursor)[i0 >> 2] = make_uint4(e0, e1, e2, e3);
      if (cursor2 && i0 < nb) reinterpret_cast<uint4*>(cursor2)[i0 >> 2] = make_uint4(e0, e1, e2, e3);
      if (chunk_bin) {              // chunks whose first position p0 = c * chunk lies in [start, end) of a non-empty bin
        const uint32_t st[5] = {e0, e1, e2, e3, e4};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (st[k + 1] == st[k]) continue;
          const uint64_t c0 = chunk_shift >= 0 ? ((uint64_t)st[k] + chunk - 1u) >> chunk_shift
                                               : ((uint64_t)st[k] + chunk - 1u) / chunk;
          for (uint64_t c = c0; c * chunk < (uint64_t)st[k + 1]; ++c) chunk_bin[c] = i0 + (uint32_t)k;
        }
      }
      carry += tile_total;
    }
  }
  if (tid == 0) status->n_valid = carry;
}

// group_offsets_kernel, the exclusive scan of the bin totals and (2-D tile order) the chunk -> first-tile table in ONE
// launch instead of three: every workgroup turns its 256 histogram columns into per-slice start offsets and publishes
// their totals; the workgroup that draws the last ticket (an agent-scope atomic in the status block, release fence
// before it, acquire fence after it) scans all nb totals into cursor[] (= bin_start[]) with coalesced 16-byte pieces --
// thread-local prefix of 4, wave scan by cross-lane shifts, four wave totals through LDS -- and, while it has every
// bin's [start, end) in registers, records for every chunk of `chunk` grouped positions the tile that holds the chunk's
// first position (what tile_chunk_bins_kernel found by bisection).  The separate scan kernel took 26 us for 16 384 bins
// (one uncoalesced serial pass per thread) and two more launch gaps; this is ~6 us in all (profiles/r05_tuning.md).
// totals / cursor must be allocated with nb rounded up to a multiple of 4 entries.
__global__ __launch_bounds__(BLOCK) void group_offsets_scan_kernel(uint32_t* hist, uint32_t nblocks, uint32_t nb,
                                                                   uint32_t* totals, uint32_t* cursor,
                                                                   StatusBlock* status, uint64_t n_pos, uint32_t chunk,
                                                                   uint32_t* chunk_bin) {
  __shared__ uint32_t s_wave[BLOCK / 64];
  __shared__ uint32_t s_last;
  const uint32_t tid = threadIdx.x;
  const uint32_t bin = blockIdx.x * BLOCK + tid;
  if (bin < nb) {
    uint32_t run = 0;
    uint32_t b = 0;
    for (; b + 8 <= nblocks; b += 8) {
      uint32_t h[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) h[u] = hist[(uint64_t)(b + u) * nb + bin];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        hist[(uint64_t)(b + u) * nb + bin] = run;
        run += h[u];
      }
    }
    for (; b < nblocks; ++b) {
      const uint32_t h = hist[(uint64_t)b * nb + bin];
      hist[(uint64_t)b * nb + bin] = run;
      run += h;
    }
    totals[bin] = run;
  }
  __threadfence();                 // the totals of this workgroup are visible device-wide before its ticket is
  __syncthreads();
  if (tid == 0) s_last = (atomicAdd(&status->ticket, 1u) == gridDim.x - 1u) ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;             // (workgroup-uniform)
  __threadfence();                 // every other workgroup's totals are visible to this one
  scan_bin_totals<BLOCK>(totals, nb, cursor, nullptr, status, chunk, chunk_bin, s_wave);
}

template <class T>
__global__ __launch_bounds__(BLOCK) void group_scatter_kernel(const uint32_t* idx, const T* sval, uint64_t nq,
                                                              uint64_t slice, const uint32_t* slice_off,
                                                              const uint32_t* base, uint32_t nb, uint4* rec) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem_raw);
  const uint32_t* off = slice_off + (uint64_t)blockIdx.x * nb;
  for (uint32_t i = threadIdx.x; i < nb; i += BLOCK) cur[i] = off[i] + base[i];
  __syncthreads();
  const uint64_t q_begin = (uint64_t)blockIdx.x * slice;
  uint64_t q_end = q_begin + slice;
  if (q_end > nq) q_end = nq;
  for (uint64_t qi = q_begin + threadIdx.x; qi < q_end; qi += BLOCK) {
    const uint32_t i = NDI_CHK(idx[qi], nb, BC_BIN);
    const uint32_t pos = NDI_CHK(atomicAdd(&cur[i], 1u), nq, BC_POSITION);
    rec[pos] = make_rec<T>((uint32_t)qi, i, sval[qi]);
  }
}

// One workgroup streams CQ grouped queries x one 256*U-vector row segment.  The four operand
// row segments stay in registers while the interval does not change, so table traffic is
// ~ (1/CQ + 1/queries-per-interval) of the gather formulation and the kernel is an output stream.
// FULL: the row length is a whole number of 256*U-vector segments, so no per-vector bounds test is needed and
// the loop body is straight-line code.  That matters for more than the branches: with control flow between the
// (conditional) table loads and their first use the compiler cannot count outstanding memory operations and
// drains them all (`s_waitcnt vmcnt(0)`, which on CDNA4 includes the wave's own stores) before every one of the
// U store groups -- one 1 KiB store in flight per wave.  Straight-line, it waits for exactly the loads a segment
// needs and the stores of several queries stay in flight.
template <class T, int STRAT, int U, int CQ, bool NT = true, bool FULL = false>
__global__ __launch_bounds__(BLOCK) void eval_bucketed_kernel(Eval1Args<T> A) {
  constexpr int VN = Wide<T>::N;
  using V = typename VecT<T, VN>::type;
  __shared__ uint32_t s_q[CQ];
  __shared__ uint32_t s_i[CQ];
  __shared__ T s_s[CQ];  // cubic: t          linear: raw x
  const uint64_t LV = A.lanes / VN;
  const uint32_t seg_vecs = BLOCK * U;
  const uint32_t segs = (uint32_t)((LV + seg_vecs - 1) / seg_vecs);
  const unsigned long long n_valid = A.nq;  // every query is grouped ...
  unsigned long long limit = A.status->first_fail[0];  // ... and rows at or after the first failure are skipped
  if (limit > A.nq) limit = A.nq;
  const uint64_t nchunks = (n_valid + CQ - 1) / CQ;
  // XCD-aware chunk order: workgroups b and b+8 share an XCD (and its L2), so XCD k walks the contiguous
  // chunk range [k*per, (k+1)*per): neighbouring chunks belong to the same or adjacent intervals and
  // re-use each other's operand rows from that L2 instead of fetching them once per XCD.
  // A workgroup takes a RUN of A.run consecutive chunks and keeps the operand rows in registers across the
  // chunk boundaries, so a table row is fetched about once per interval instead of once per chunk it appears in:
  // table reads interleaved into the write stream cost more than their share of the bytes (tools/tune_bucketed.hip:
  // 4 operand rows per chunk 4.90 ms, 1 row 4.68 ms, none 4.68 ms = the store-only ceiling).
  // (Placement is a speed heuristic only -- any mapping gives the same result.)
  const uint64_t per = (nchunks + 7) / 8;
  const uint64_t run = A.run ? A.run : 1;
  const uint64_t runs_per_xcd = (per + run - 1) / run;
  for (uint32_t seg = blockIdx.y; seg < segs; seg += gridDim.y) {
    const uint64_t v0 = (uint64_t)seg * seg_vecs + threadIdx.x;
    for (uint64_t vb = blockIdx.x; vb < runs_per_xcd * 8; vb += gridDim.x) {
      const uint64_t xcd = vb & 7u;
      const uint64_t c_begin = xcd * per + (vb >> 3) * run;
      uint64_t c_end = c_begin + run;
      if (c_end > (xcd + 1) * per) c_end = (xcd + 1) * per;
      if (c_end > nchunks) c_end = nchunks;
      V ryl[U], ryr[U], ra[U], rb[U];
      uint32_t cur = 0xffffffffu;
      for (uint64_t chunk = c_begin; chunk < c_end; ++chunk) {
        const uint64_t p0 = chunk * CQ;
        const uint32_t cnt = (n_valid - p0 < (uint64_t)CQ) ? (uint32_t)(n_valid - p0) : (uint32_t)CQ;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < cnt; j += BLOCK) {
          const uint4 r = A.rec[p0 + j];
          s_q[j] = r.x;
          s_i[j] = r.y;
          s_s[j] = rec_value(r, T(0));
        }
        __syncthreads();
        auto emit = [&](uint32_t i, uint32_t qi, T sj) {
          const RowCoef<T, STRAT> c = row_coef<T, STRAT>(A.knots, i, sj, sj);
          V* o = reinterpret_cast<V*>(A.out + (uint64_t)qi * A.out_stride);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint64_t v = v0 + (uint64_t)u * BLOCK;
            if (FULL || v < LV) store_stream<NT>(o + v, row_point<T, STRAT, V>(c, ryl[u], ryr[u], ra[u], rb[u]));
          }
        };
        for (uint32_t j = 0; j < cnt;) {
          const uint32_t i = NDI_CHK((uint32_t)__builtin_amdgcn_readfirstlane((int)s_i[j]), A.n_int, BC_INTERVAL);
          const uint32_t qi = NDI_CHK((uint32_t)__builtin_amdgcn_readfirstlane((int)s_q[j]), A.nq, BC_QUERY);
          if (qi >= limit) { ++j; continue; }
          if (i != cur) {
            cur = i;
            const V* yl = reinterpret_cast<const V*>(A.data + (uint64_t)i * A.lanes);
            const V* yr = yl + LV;
            const V* pa = reinterpret_cast<const V*>(A.ca + (uint64_t)i * A.lanes);
            const V* pb = reinterpret_cast<const V*>(A.cb + (uint64_t)i * A.lanes);
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const uint64_t v = v0 + (uint64_t)u * BLOCK;
              if (FULL || v < LV) {
                ryl[u] = yl[v];
                ryr[u] = yr[v];
                if (STRAT == ST_CUBIC) {
                  ra[u] = pa[v];
                  rb[u] = pb[v];
                }
              }
            }
            // NOTE (measured, tools/_run7 / profiles/r02_tuning.md): draining the loads here with an explicit
            // s_waitcnt vmcnt(0), so that the compute/store section below carries no waits at all and the stores run
            // ahead without limit, is SLOWER (5.10-5.37 ms vs 4.93): the waits the compiler places below for the
            // reload path (vmcnt(28) ... vmcnt(7)) act on the wave's own stores on the common path -- vmcnt counts
            // stores on CDNA4 -- and keep about one query's stores in flight per wave, which is the pacing the
            // memory system runs best at together with the polynomial's VALU work.  (Round 5, profiles/r05_tuning.md 6:
            // the same-interval queries in a loop of their own with an EXPLICIT s_waitcnt vmcnt(N), N = 4 / 8 / 12 / 20
            // stores in flight per wave -- every N within the run-to-run noise of the implicit pacing, f64 and f32.)
          }
          emit(i, qi, s_s[j]);
          ++j;
        }
      }
    }
  }
}

// BUCKETED for rows shorter than one workgroup pass (fewer than 256 vectors): the workgroup is split into NG = 256 / G
// groups of G threads (G = power of two >= vectors per row; lanes beyond the row idle), each group streams its own
// run of CQ consecutive grouped records and keeps the interval's four operand vectors in registers -- one vector per
// thread and table -- while the interval does not change.  The grouped records of a workgroup pass (NG * CQ of them)
// are staged in LDS with one coalesced load.  As in eval_bucketed_kernel the table traffic all but disappears and the
// kernel is a stream of whole output rows written at their original positions; it pays from rows of a few hundred
// bytes upwards (shorter rows are partial-line scattered writes: the query-order kernel above is the one for them).
template <class T, int STRAT, int G, int CQ, bool NT = true>
__global__ __launch_bounds__(BLOCK) void eval_bucketed_short_kernel(Eval1Args<T> A) {
  constexpr int VN = Wide<T>::N;
  using V = typename VecT<T, VN>::type;
  constexpr uint32_t NG = BLOCK / G;
  constexpr uint32_t WQ = NG * CQ;
  __shared__ uint32_t s_q[WQ];
  __shared__ uint32_t s_i[WQ];
  __shared__ T s_s[WQ];  // cubic: t          linear: raw x
  const uint32_t LV = (uint32_t)(A.lanes / VN);
  const uint32_t g = threadIdx.x / G, v = threadIdx.x % G;
  const bool vlive = v < LV;
  const unsigned long long n_valid = A.nq;             // every query is grouped ...
  unsigned long long limit = A.status->first_fail[0];  // ... and rows at or after the first failure are skipped
  if (limit > A.nq) limit = A.nq;
  const uint64_t nchunks = (n_valid + WQ - 1) / WQ;
  const uint64_t per = (nchunks + 7) / 8;              // XCD-aware chunk order as in eval_bucketed_kernel
  V ryl = V(0), ryr = V(0), ra = V(0), rb = V(0);
  uint32_t cur = 0xffffffffu;
  for (uint64_t vb = blockIdx.x; vb < per * 8; vb += gridDim.x) {
    const uint64_t chunk = (vb & 7u) * per + (vb >> 3);
    if (chunk >= nchunks) continue;                    // (workgroup-uniform)
    const uint64_t p0 = chunk * WQ;
    const uint32_t cnt = (n_valid - p0 < (uint64_t)WQ) ? (uint32_t)(n_valid - p0) : WQ;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < cnt; j += BLOCK) {
      const uint4 r = A.rec[p0 + j];
      s_q[j] = r.x;
      s_i[j] = r.y;
      s_s[j] = rec_value(r, T(0));
    }
    __syncthreads();
    const uint32_t j0 = g * CQ;
    const uint32_t j1 = (j0 + CQ < cnt) ? j0 + CQ : cnt;
    for (uint32_t j = j0; j < j1; ++j) {
      const uint32_t i = NDI_CHK(s_i[j], A.n_int, BC_INTERVAL);
      const uint32_t qi = NDI_CHK(s_q[j], A.nq, BC_QUERY);
      if (qi >= limit) continue;
      const T sj = s_s[j];
      if (i != cur) {
        cur = i;
        if (vlive) {
          const V* yl = reinterpret_cast<const V*>(A.data + (uint64_t)i * A.lanes);
          ryl = yl[v];
          ryr = yl[LV + v];
          if (STRAT == ST_CUBIC) {
            ra = reinterpret_cast<const V*>(A.ca + (uint64_t)i * A.lanes)[v];
            rb = reinterpret_cast<const V*>(A.cb + (uint64_t)i * A.lanes)[v];
          }
        }
      }
      const RowCoef<T, STRAT> c = row_coef<T, STRAT>(A.knots, i, sj, sj);
      V* o = reinterpret_cast<V*>(A.out + (uint64_t)qi * A.out_stride);
      if (vlive) store_stream<NT>(o + v, row_point<T, STRAT, V>(c, ryl, ryr, ra, rb));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 2-D bilinear (bilinear.rs:83-97): z = frac(y; frac(x; z11,z21), frac(x; z12,z22))
// ---------------------------------------------------------------------------------------------
template <class T>
struct Eval2Args {
  const T* xk;
  const T* yk;
  const T* data;  // [nx][ny][lanes]
  const T* qx;
  const T* qy;
  const uint32_t* xi;
  const uint32_t* yi;
  T* out;
  uint64_t nx, ny, lanes, out_stride, nq;
  // addressing of the corner rows: element offset of cell (xi, yi) = (xi * row_cells + yi) * cell_elems.
  // plain layout [nx][ny][lanes]: row_cells = ny, cell_elems = lanes; pair-packed layout (see
  // pack_pairs_kernel): row_cells = ny - 1, cell_elems = 2 * lanes.  z12 is always z11 + lanes.
  uint64_t row_cells, cell_elems;
  const StatusBlock* status;
  // tile-grouped order (nullptr: query order): per grouped position the query's record (group_scatter2d_kernel)
  const uint4* rec_i;    // {query index, xi, yi, 0}, or the compact {query index, xi | yi << 16, qx, qy} (f32)
  const T* rec_q;        // {qx, qy} pairs; nullptr with compact records
  // eval_bilinear_tiles_kernel: bin_start[b] = first grouped position of tile b (exclusive scan of the tile counts),
  // tiles of 2^ts x 2^ts cells, ntx x nty of them, `chunk` grouped positions per unit of work
  const uint32_t* bin_start;
  uint32_t nb, ts, nty, chunk;
  const uint32_t* chunk_bin;   // [chunks] the tile that holds each chunk's first grouped position (tile_chunk_bins_kernel)
  int debug;                   // NDI_TUNING builds only: bit 2 = no stores (measurement aid)
  // ceil(2^32 / (cols * LV)) for a full tile row (2^ts + 1 grid points) and for the last tile column's shorter rows:
  // item -> (grid row, vector) of the tile staging by one v_mul_hi instead of an emulated division
  uint32_t rvm_full, rvm_edge;
  uint32_t ch_split;           // eval_bilinear_tiles_kernel: 0 / 1 = every workgroup handles whole rows, 2 = one half each
};

template <class T, bool LDS>
struct KnotPtr { using type = const T*; };
template <class T>
struct KnotPtr<T, true> { using type = const __attribute__((address_space(3))) T*; };

// 2-D grouping (tile-grouped order): locate2_kernel leaves one tile histogram per query slice (the key of a query
// is the tile of 2^sx x 2^sy cells its cell falls in); group_offsets_kernel / bucket_scan_kernel turn them into
// start offsets as in the 1-D case; this kernel re-reads each slice and places every query's record at its grouped
// position with block-local LDS cursors.  COMPACT (f32, both axes < 65536 knots): one self-contained 16-byte record
// {query index, xi | yi << 16, bits(qx), bits(qy)} -- the evaluation then reads ONE sequential 16-byte stream and
// never touches xi / yi / qx / qy by query index; otherwise {query index, xi, yi} + a {qx, qy} pair per position.
template <class T, bool COMPACT>
__global__ __launch_bounds__(1024) void group_scatter2d_kernel(const uint32_t* xi, const uint32_t* yi, const T* qx,
                                                                const T* qy, uint64_t nq, uint64_t slice,
                                                                const uint32_t* slice_off, const uint32_t* base,
                                                                uint32_t nb, uint32_t sx, uint32_t sy, uint32_t nty,
                                                                uint4* rec_i, T* rec_q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem_raw);
  // (Measured at C3 and dropped, profiles/r04_tuning.md: an XCD-aware slice order -- a line's writers on one L2 -- equal;
  // one cursor row per XCD drawn with device-scope atomics, so that a tile's records arrive as 8 sequential streams -- 2 x
  // slower, the atomics are served beyond L2; non-temporal record stores -- +0.2 ms, the partial lines do merge in L2.)
  const uint32_t sl = blockIdx.x;
  const uint32_t* off = slice_off + (uint64_t)sl * nb;
  for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) cur[i] = off[i] + base[i];
  __syncthreads();
  const uint64_t q_begin = (uint64_t)sl * slice;
  uint64_t q_end = q_begin + slice;
  if (q_end > nq) q_end = nq;
  for (uint64_t qi = q_begin + threadIdx.x; qi < q_end; qi += blockDim.x) {
    uint32_t ix = xi[qi], iy;
    if (yi) iy = yi[qi];
    else { iy = ix >> 16; ix &= 0xffffu; }   // locate2_kernel's cell word
    const T x = qx[qi], y = qy[qi];
    const uint32_t pos = NDI_CHK(atomicAdd(&cur[NDI_CHK((ix >> sx) * nty + (iy >> sy), nb, BC_BIN)], 1u), nq, BC_POSITION);
    if constexpr (COMPACT && sizeof(T) == 4) {
      const uint4 r = make_uint4((uint32_t)qi, ix | (iy << 16), __builtin_bit_cast(uint32_t, x),
                                 __builtin_bit_cast(uint32_t, y));
      rec_i[pos] = r;   // (non-temporal: +0.2 ms at C3 -- the records of a line merge in L2)
    } else {
      rec_i[pos] = make_uint4((uint32_t)qi, ix, iy, 0u);
      rec_q[2 * (uint64_t)pos] = x;
      rec_q[2 * (uint64_t)pos + 1] = y;
    }
  }
}

// Two-level 2-D grouping.  The one-pass scatter above places 2.4 records per (slice, tile) at C3: every 16-byte record
// is a partial line whose other records come from other slices -- other XCDs, other L2s -- so each reaches memory as
// its own 32-byte sector (PMC: 304 MiB written for 153 MiB of records) at the rate of random memory transactions
// (profiles/r05_tuning.md 2).  Two passes whose runs are whole lines instead:
//   coarse_scatter2d_kernel  slice -> tile ROW (ntx bins): a slice leaves ~300 consecutive records per row, all written
//                            by one workgroup -- the partial lines merge in that workgroup's L2;
//   fine_scatter2d_kernel    tile row -> tile (nty bins), `round` records of a row at a time: LDS ranks, ONE global
//                            atomic per (round, tile) claims the run's place behind the tile's cursor (the order of
//                            the records inside a tile is free: every record writes its own output row), then the
//                            records of a run are written next to each other by one workgroup.
// No launch between locate2_kernel and the coarse pass: every coarse workgroup forms the start of its slice in every
// tile row from locate2_kernel's row histograms chist[slice][ntx] (a 128 KiB table at C3, read from L2: the rows of
// the slices before it, and all of them for the row totals), and -- its loads in flight beside the latency-bound
// scatter -- the tile totals of ITS share of the tile columns of the per-slice tile histograms (column-block-major,
// Locate2Args::hist_w); scan_bin_totals_kernel (one workgroup) then scans the nb totals into bin_start[] / cursor2[] /
// chunk_bin[], which the fine pass and the evaluation read.  (The scan by the coarse workgroup that draws the last
// ticket was measured: the agent-scope release fence in front of the ticket writes the XCD's L2 back -- an L2 full of
// partly written record lines -- 256 times: 0.33 ms behind the scatter, 0.19 ms in front of it, against 0.08 + 0.01.)
template <class T, bool COMPACT, bool SORT = false>
__global__ __launch_bounds__(1024) void coarse_scatter2d_kernel(const uint32_t* xi, const uint32_t* yi, const T* qx,
                                                                 const T* qy, uint64_t nq, uint64_t slice,
                                                                 const uint32_t* chist, const uint32_t* hist, uint32_t nb,
                                                                 uint32_t ntx, uint32_t sx, uint4* rec_i, T* rec_q,
                                                                 uint32_t* totals) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ uint32_t s_wave[16];
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem_raw);   // [ntx]
  uint32_t* part_b = cur + ntx;                            // [parts][ntx]: rows of the slices before this one
  uint32_t* part_a = part_b + blockDim.x + ntx;            // [parts][ntx]: rows of all slices; later [blockDim.x] column sums
  const uint32_t sl = blockIdx.x, tid = threadIdx.x, P = gridDim.x;
  const uint32_t parts = blockDim.x >= ntx ? blockDim.x / ntx : 1u;
  const uint32_t per = (P + parts - 1u) / parts;
  for (uint32_t item = tid; item < parts * ntx; item += blockDim.x) {
    const uint32_t p = item / ntx, c = item - p * ntx;
    const uint32_t s0 = p * per, s1 = (s0 + per < P) ? s0 + per : P;
    uint32_t before = 0, all = 0;
    uint32_t s = s0;
    for (; s + 8u <= s1; s += 8u) {
      uint32_t h[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) h[u] = chist[(uint64_t)(s + u) * ntx + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        all += h[u];
        if (s + (uint32_t)u < sl) before += h[u];
      }
    }
    for (; s < s1; ++s) {
      const uint32_t h = chist[(uint64_t)s * ntx + c];
      all += h;
      if (s < sl) before += h;
    }
    part_b[item] = before;
    part_a[item] = all;
  }
  __syncthreads();
  // exclusive scan of the row totals, blockDim.x rows at a time (wave scan by cross-lane shifts, wave totals through LDS)
  {
    const uint32_t lane = tid & 63u, wave = tid >> 6, nwaves = (blockDim.x + 63u) >> 6;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < ntx; c0 += blockDim.x) {
      const uint32_t c = c0 + tid;
      uint32_t before = 0, all = 0;
      if (c < ntx)
        for (uint32_t p = 0; p < parts; ++p) { before += part_b[p * ntx + c]; all += part_a[p * ntx + c]; }
      uint32_t incl = all;
#pragma unroll
      for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += up;
      }
      __syncthreads();
      if (lane == 63u) s_wave[wave] = incl;
      __syncthreads();
      uint32_t wave_off = 0, total = 0;
      for (uint32_t w = 0; w < nwaves; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wave) wave_off += t;
        total += t;
      }
      if (c < ntx) cur[c] = carry + wave_off + incl - all + before;
      carry += total;
    }
  }
  __syncthreads();   // (part_a is free from here on)
  // this workgroup's share of the tile totals: columns [col0, col0 + cpw) of hist[P][nb], `w` columns at a time by
  // blockDim.x / w row groups, reduced through LDS
  {
    const uint32_t cpw = (nb + P - 1u) / P;
    const uint32_t col0 = sl * cpw;
    const uint32_t w = cpw < blockDim.x ? cpw : blockDim.x;
    const uint32_t nrg = blockDim.x / w;
    for (uint32_t cb = 0; cb < cpw; cb += w) {
      const uint32_t col = tid % w, rg = tid / w;
      const uint32_t bin = col0 + cb + col;
      uint32_t sum = 0;
      if (rg < nrg && cb + col < cpw && bin < nb) {
        const uint32_t* h = hist + (uint64_t)sl * P * cpw + cb + col;   // column-block-major (Locate2Args::hist_w = cpw)
        uint32_t r = rg;
        for (; r + 3u * nrg < P; r += 4u * nrg) {
          const uint32_t h0 = h[(uint64_t)r * cpw], h1 = h[(uint64_t)(r + nrg) * cpw];
          const uint32_t h2 = h[(uint64_t)(r + 2u * nrg) * cpw], h3 = h[(uint64_t)(r + 3u * nrg) * cpw];
          sum += h0 + h1 + h2 + h3;
        }
        for (; r < P; r += nrg) sum += h[(uint64_t)r * cpw];
      }
      __syncthreads();
      part_a[tid] = sum;
      __syncthreads();
      if (tid < w && cb + tid < cpw && col0 + cb + tid < nb) {
        uint32_t t = 0;
        for (uint32_t g = 0; g < nrg; ++g) t += part_a[g * w + tid];
        totals[col0 + cb + tid] = t;
      }
    }
  }
  const uint64_t q_begin = (uint64_t)sl * slice;
  uint64_t q_end = q_begin + slice;
  if (q_end > nq) q_end = nq;
  constexpr int U = 4;   // queries per thread and trip; the next trip's loads are in flight while this one's records are
                         // placed (the chain load -> LDS atomic -> store is latency)
  if (q_begin >= q_end) return;
  uint32_t nix[U], niy[U];
  T nx_[U], ny_[U];
  if constexpr (SORT && COMPACT && sizeof(T) == 4) {
    // SORTED rounds (compact f32 records): the U * blockDim.x records of a trip are ordered by tile row in LDS -- ranks by
    // LDS atomics, an exclusive scan of the trip's row counts, every row's run placed behind the slice's cursor -- and
    // copied out by neighbouring lanes, four 16-byte records to a 64-byte request, instead of one request per record
    // (1e7 of them at C3: a third of the pass, profiles/r05_tuning.md 2).
    // LDS behind [cur | part_b | part_a]: [s_cnt | s_off | s_base : ntx each | records U * blockDim.x | destinations]
    uint32_t* s_cnt = part_a + blockDim.x + ntx;
    uint32_t* s_off = s_cnt + ntx;
    uint32_t* s_base = s_off + ntx;
    uint4* s_rec = reinterpret_cast<uint4*>(smem_raw + ((((size_t)(6u * ntx + 2u * blockDim.x)) * 4u + 15u) & ~(size_t)15u));
    uint32_t* s_dst = reinterpret_cast<uint32_t*>(s_rec + (size_t)U * blockDim.x);
    const uint32_t lane = tid & 63u, wave = tid >> 6, nwaves = (blockDim.x + 63u) >> 6;
    for (uint32_t b = tid; b < ntx; b += blockDim.x) s_cnt[b] = 0u;
    __syncthreads();
    auto fetch2 = [&](uint64_t q0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t qi = q0 + (uint64_t)u * blockDim.x;
        const uint64_t qc = qi < q_end ? qi : q_end - 1u;
        nix[u] = xi[qc];
        niy[u] = yi ? yi[qc] : 0u;
        nx_[u] = qx[qc];
        ny_[u] = qy[qc];
      }
    };
    fetch2(q_begin + tid);
    for (uint64_t qb = q_begin; qb < q_end; qb += (uint64_t)U * blockDim.x) {   // (workgroup-uniform)
      uint4 rec[U];
      uint32_t row[U], rank[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t qi = qb + (uint64_t)u * blockDim.x + tid;
        uint32_t jx = nix[u], jy = niy[u];
        if (!yi) { jy = jx >> 16; jx &= 0xffffu; }
        rec[u] = make_uint4((uint32_t)qi, jx | (jy << 16), __builtin_bit_cast(uint32_t, nx_[u]), __builtin_bit_cast(uint32_t, ny_[u]));
        row[u] = NDI_CHK(jx >> sx, ntx, BC_BIN);
      }
      fetch2(qb + (uint64_t)U * blockDim.x + tid);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t qi = qb + (uint64_t)u * blockDim.x + tid;
        rank[u] = qi < q_end ? atomicAdd(&s_cnt[row[u]], 1u) : 0u;
      }
      __syncthreads();
      uint32_t carry = 0;
      for (uint32_t b0 = 0; b0 < ntx; b0 += blockDim.x) {
        const uint32_t b = b0 + tid;
        const uint32_t n = b < ntx ? s_cnt[b] : 0u;
        uint32_t incl = n;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
          const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
          if (lane >= d) incl += up;
        }
        __syncthreads();
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
        for (uint32_t w = 0; w < nwaves; ++w) {
          const uint32_t t = s_wave[w];
          if (w < wave) wave_off += t;
          total += t;
        }
        if (b < ntx) {
          s_off[b] = carry + wave_off + incl - n;
          s_base[b] = cur[b];
          cur[b] += n;
          s_cnt[b] = 0u;
        }
        carry += total;
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t qi = qb + (uint64_t)u * blockDim.x + tid;
        if (qi < q_end) {
          const uint32_t i = NDI_CHK(s_off[row[u]] + rank[u], (uint32_t)U * blockDim.x, BC_POSITION);
          s_rec[i] = rec[u];
          s_dst[i] = s_base[row[u]] + rank[u];
        }
      }
      __syncthreads();
      const uint64_t left = q_end - qb;
      const uint32_t n_here = left < (uint64_t)U * blockDim.x ? (uint32_t)left : (uint32_t)U * blockDim.x;
      for (uint32_t i = tid; i < n_here; i += blockDim.x) rec_i[NDI_CHK(s_dst[i], nq, BC_POSITION)] = s_rec[i];
    }
    return;
  }
  auto fetch = [&](uint64_t q0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t qi = q0 + (uint64_t)u * blockDim.x;
      const uint64_t qc = qi < q_end ? qi : q_end - 1u;
      nix[u] = xi[qc];
      niy[u] = yi ? yi[qc] : 0u;
      nx_[u] = qx[qc];
      ny_[u] = qy[qc];
    }
  };
  fetch(q_begin + tid);
  for (uint64_t q0 = q_begin + tid; q0 < q_end; q0 += (uint64_t)U * blockDim.x) {
    uint32_t ix[U], iy[U];
    T x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { ix[u] = nix[u]; iy[u] = niy[u]; x[u] = nx_[u]; y[u] = ny_[u]; }
    fetch(q0 + (uint64_t)U * blockDim.x);     // (clamped: past the slice it re-reads the slice's last query)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t qi = q0 + (uint64_t)u * blockDim.x;
      if (qi >= q_end) break;
      uint32_t jx = ix[u], jy = iy[u];
      if (!yi) { jy = jx >> 16; jx &= 0xffffu; }   // locate2_kernel's cell word
      const uint32_t pos = NDI_CHK(atomicAdd(&cur[NDI_CHK(jx >> sx, ntx, BC_BIN)], 1u), nq, BC_POSITION);
      if constexpr (COMPACT && sizeof(T) == 4) {
        rec_i[pos] = make_uint4((uint32_t)qi, jx | (jy << 16), __builtin_bit_cast(uint32_t, x[u]),
                                __builtin_bit_cast(uint32_t, y[u]));
      } else {
        rec_i[pos] = make_uint4((uint32_t)qi, jx, jy, 0u);
        rec_q[2 * (uint64_t)pos] = x[u];
        rec_q[2 * (uint64_t)pos + 1] = y[u];
      }
    }
  }
}

__global__ __launch_bounds__(1024) void scan_bin_totals_kernel(const uint32_t* totals, uint32_t nb, uint32_t* cursor,
                                                               uint32_t* cursor2, StatusBlock* status, uint32_t chunk,
                                                               uint32_t* chunk_bin) {
  __shared__ uint32_t s_wave[16];
  scan_bin_totals<1024>(totals, nb, cursor, cursor2, status, chunk, chunk_bin, s_wave);
}

// Workgroup (row c, part g of G) takes the rounds g, g + G, ... of `R * blockDim.x` records of tile row c's run
// [bin_start[c * nty], bin_start[(c + 1) * nty]) of the coarse order; any G is correct (the host sizes it for about
// one round per workgroup on evenly spread queries).  When the grid is a multiple of 8 * G the parts of a row sit on
// one XCD (blockIdx.x round-robins over the XCDs): the lines at the seams of neighbouring rounds' runs merge in one L2.
template <class T, bool COMPACT, int R>
__global__ __launch_bounds__(1024) void fine_scatter2d_kernel(const uint4* in_i, const T* in_q, uint4* out_i, T* out_q,
                                                               const uint32_t* bin_start, uint32_t* cursor2,
                                                               uint64_t nq, uint32_t ntx, uint32_t nty, uint32_t sy,
                                                               uint32_t G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem_raw);   // [nty]
  uint32_t* s_base = s_cnt + nty;                            // [nty]
  const uint32_t tid = threadIdx.x;
  uint32_t c, g;
  if ((gridDim.x % (8u * G)) == 0u) {
    const uint32_t x = blockIdx.x & 7u, r = blockIdx.x >> 3;
    g = r % G;
    c = (r / G) * 8u + x;
  } else {
    c = blockIdx.x / G;
    g = blockIdx.x - c * G;
  }
  if (c >= ntx) return;
  const uint64_t p_begin = bin_start[(uint64_t)c * nty];
  const uint64_t p_end = (c + 1u < ntx) ? (uint64_t)bin_start[(uint64_t)(c + 1u) * nty] : nq;
  for (uint32_t f = tid; f < nty; f += blockDim.x) s_cnt[f] = 0u;
  __syncthreads();
  const uint64_t round = (uint64_t)R * blockDim.x;
  uint64_t r0 = p_begin + (uint64_t)g * round;
  if (r0 >= p_end) return;         // (workgroup-uniform)
  // the records of the next round are in flight while this round's ranks are drawn, its runs claimed and written
  uint4 nrec[R];
  T nrx[R], nry[R];
  auto fetch = [&](uint64_t rr) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint64_t p = rr + (uint64_t)k * blockDim.x + tid;
      const uint64_t pc = p < p_end ? p : p_end - 1u;
      nrec[k] = in_i[pc];
      if constexpr (!(COMPACT && sizeof(T) == 4)) { nrx[k] = in_q[2 * pc]; nry[k] = in_q[2 * pc + 1]; }
    }
  };
  fetch(r0);
  for (; r0 < p_end; r0 += (uint64_t)G * round) {
    uint4 rec[R];
    T rx[R], ry[R];
    uint32_t f[R], rank[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      rec[k] = nrec[k];
      if constexpr (!(COMPACT && sizeof(T) == 4)) { rx[k] = nrx[k]; ry[k] = nry[k]; }
    }
    if (r0 + (uint64_t)G * round < p_end) fetch(r0 + (uint64_t)G * round);   // (workgroup-uniform)
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint64_t p = r0 + (uint64_t)k * blockDim.x + tid;
      const uint32_t jy = (COMPACT && sizeof(T) == 4) ? rec[k].y >> 16 : rec[k].z;
      f[k] = NDI_CHK(jy >> sy, nty, BC_BIN);
      rank[k] = p < p_end ? atomicAdd(&s_cnt[f[k]], 1u) : 0u;
    }
    __syncthreads();
    for (uint32_t b = tid; b < nty; b += blockDim.x) {
      const uint32_t n = s_cnt[b];
      s_base[b] = n ? atomicAdd(&cursor2[(uint64_t)c * nty + b], n) : 0u;
      s_cnt[b] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint64_t p = r0 + (uint64_t)k * blockDim.x + tid;
      if (p >= p_end) break;
      const uint64_t pos = NDI_CHK(s_base[f[k]] + rank[k], nq, BC_POSITION);
      out_i[pos] = rec[k];
      if constexpr (!(COMPACT && sizeof(T) == 4)) { out_q[2 * pos] = rx[k]; out_q[2 * pos + 1] = ry[k]; }
    }
  }
}

// fine_scatter2d_kernel with the round's records SORTED in LDS before they leave (compact f32 records): the direct form
// writes every 16-byte record as a request of its own (64 per store instruction); here a run's records are copied out by
// neighbouring lanes, four to a 64-byte request.  [s_cnt | s_base | s_off | wave totals | records | destinations]
template <int R, int TB>
__global__ __launch_bounds__(TB) void fine_scatter2d_sorted_kernel(const uint4* in_i, uint4* out_i, const uint32_t* bin_start,
                                                                   uint32_t* cursor2, uint64_t nq, uint32_t ntx,
                                                                   uint32_t nty, uint32_t sy, uint32_t G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem_raw);   // [nty]
  uint32_t* s_base = s_cnt + nty;                            // [nty]
  uint32_t* s_off = s_base + nty;                            // [nty]
  uint32_t* s_wave = s_off + nty;                            // [16]
  uint4* s_rec = reinterpret_cast<uint4*>(smem_raw + (((size_t)(3u * nty + 16u) * 4u + 15u) & ~(size_t)15u));   // [R * TB]
  uint32_t* s_dst = reinterpret_cast<uint32_t*>(s_rec + (size_t)R * TB);                                         // [R * TB]
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t c, g;
  if ((gridDim.x % (8u * G)) == 0u) {
    const uint32_t x = blockIdx.x & 7u, r = blockIdx.x >> 3;
    g = r % G;
    c = (r / G) * 8u + x;
  } else {
    c = blockIdx.x / G;
    g = blockIdx.x - c * G;
  }
  if (c >= ntx) return;
  const uint64_t p_begin = bin_start[(uint64_t)c * nty];
  const uint64_t p_end = (c + 1u < ntx) ? (uint64_t)bin_start[(uint64_t)(c + 1u) * nty] : nq;
  for (uint32_t f = tid; f < nty; f += TB) s_cnt[f] = 0u;
  __syncthreads();
  const uint64_t round = (uint64_t)R * TB;
  for (uint64_t r0 = p_begin + (uint64_t)g * round; r0 < p_end; r0 += (uint64_t)G * round) {
    uint4 rec[R];
    uint32_t f[R], rank[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint64_t p = r0 + (uint64_t)k * TB + tid;
      rec[k] = in_i[p < p_end ? p : p_end - 1u];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint64_t p = r0 + (uint64_t)k * TB + tid;
      f[k] = NDI_CHK((rec[k].y >> 16) >> sy, nty, BC_BIN);
      rank[k] = p < p_end ? atomicAdd(&s_cnt[f[k]], 1u) : 0u;
    }
    __syncthreads();
    // exclusive scan of the round's tile counts (TB bins at a time), the runs' places claimed on the way
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nty; b0 += TB) {
      const uint32_t b = b0 + tid;
      const uint32_t n = b < nty ? s_cnt[b] : 0u;
      uint32_t incl = n;
#pragma unroll
      for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += up;
      }
      __syncthreads();
      if (lane == 63u) s_wave[wave] = incl;
      __syncthreads();
      uint32_t wave_off = 0, total = 0;
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)TB / 64u; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wave) wave_off += t;
        total += t;
      }
      if (b < nty) {
        s_off[b] = carry + wave_off + incl - n;
        s_base[b] = n ? atomicAdd(&cursor2[(uint64_t)c * nty + b], n) : 0u;
        s_cnt[b] = 0u;
      }
      carry += total;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint64_t p = r0 + (uint64_t)k * TB + tid;
      if (p < p_end) {
        const uint32_t i = NDI_CHK(s_off[f[k]] + rank[k], (uint32_t)round, BC_POSITION);
        s_rec[i] = rec[k];
        s_dst[i] = s_base[f[k]] + rank[k];
      }
    }
    __syncthreads();
    const uint32_t n_here = (p_end - r0 < round) ? (uint32_t)(p_end - r0) : (uint32_t)round;
    for (uint32_t i = tid; i < n_here; i += TB) out_i[NDI_CHK(s_dst[i], nq, BC_POSITION)] = s_rec[i];
  }
}

// For every chunk of `chunk` grouped positions the tile that holds its first position: the last b with
// bin_start[b] <= c * chunk (bin_start[0] = 0).  One thread per chunk, part of the grouping stage.
__global__ __launch_bounds__(BLOCK) void tile_chunk_bins_kernel(const uint32_t* bin_start, uint32_t nb, uint64_t n_pos,
                                                                uint32_t chunk, uint32_t* chunk_bin) {
  const uint64_t nchunks = (n_pos + chunk - 1) / chunk;
  const uint64_t c = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (c >= nchunks) return;
  const uint64_t p0 = c * chunk;
  uint32_t lo = 0, hi = nb;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((uint64_t)bin_start[mid] <= p0) lo = mid; else hi = mid;
  }
  chunk_bin[c] = lo;
}

// Pair-packed grid for short trailing axes: P[xi][yi] = { z[xi][yi], z[xi][yi+1] }, yi < ny-1.  The two
// corners a query needs from one grid row become one naturally aligned segment (128 B at 16 f32 channels), so a
// query touches exactly two cache lines instead of three on average (a 64-B-aligned 128-B segment straddles
// two 128-B lines half of the time).  Costs 2x the grid memory; values are copied, never recomputed.
// E = copy unit (a 16-byte vector when the cell size allows, else one element); units = copy units per grid
// point.  grid.y walks the grid rows, so the only per-unit index arithmetic is the split of the in-row position
// into (pair, unit): a shift when `units` is a power of two (unit_shift >= 0), one division otherwise.
template <class E>
__global__ __launch_bounds__(BLOCK) void pack_pairs_kernel(const E* in, E* out, uint64_t nx, uint64_t ny,
                                                           uint32_t units, int unit_shift) {
  const uint64_t row_out = (ny - 1) * 2 * units;   // copy units per packed grid row
  for (uint64_t xi = blockIdx.y; xi < nx; xi += gridDim.y) {
    const E* src = in + xi * ny * units;
    E* dst = out + xi * row_out;
    for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e < row_out; e += (uint64_t)gridDim.x * BLOCK) {
      const uint64_t pair = unit_shift >= 0 ? e >> unit_shift : e / units;   // = 2 * yi + h
      const uint64_t c = e - pair * units;
      dst[e] = src[((pair >> 1) + (pair & 1u)) * units + c];                // z[xi][yi + h][c]
    }
  }
}

// APPROX exists only for the tuning harness (tools/tune_eval.hip: how much of the kernel is division cost);
// the library instantiates the exact form.
template <class T, class V, bool APPROX = false>
__device__ __forceinline__ V frac_v(T x1, V y1, T x2, V y2, T x) {
  V m;
  if (APPROX) m = (y2 - y1) * (T(1) / (x2 - x1));
  else m = (y2 - y1) / (x2 - x1);  // Linear::calc_frac, linear.rs:33-35
  return m * (x - x1) + y1;
}

// UNR items (one output vector each) per thread and loop trip, processed in three phases -- indices and query
// values, then knots and the four corner vectors, then arithmetic and store -- so that the two dependent memory
// latencies of an item (index -> corner) are paid once per UNR items: the kernel is a random gather whose only
// lever is memory-level parallelism (C5, 8192 x 8192 x 16 f32: UNR 1 -> 2: 0.88 -> see DESIGN.md 4.4).
// KLDS: both knot vectors are staged in LDS once per workgroup (TB = 1024 threads, a grid of about two workgroups
// per CU) and the four knot values of an item come from there instead of four scattered global loads.
template <class T, int VEC, bool APPROX = false, int UNR = 2, int TB = BLOCK, bool KLDS = false, bool SDIV = false>
__global__ __launch_bounds__(TB) void eval_bilinear_kernel(Eval2Args<T> A, uint32_t tile_q) {
  using V = typename VecT<T, VEC>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typename KnotPtr<T, KLDS>::type xk, yk;
  if constexpr (KLDS) {
    T* s0 = reinterpret_cast<T*>(smem_raw);
    for (uint32_t i = threadIdx.x; i < (uint32_t)A.nx; i += TB) s0[i] = A.xk[i];
    for (uint32_t i = threadIdx.x; i < (uint32_t)A.ny; i += TB) s0[A.nx + i] = A.yk[i];
    __syncthreads();
    xk = (lds_ptr<T>)(smem_raw);
    yk = xk + A.nx;
  } else {
    xk = A.xk;
    yk = A.yk;
  }
  const uint32_t LV = (uint32_t)(A.lanes / VEC);
  unsigned long long limit = A.status->first_fail[0];
  if (A.status->first_fail[1] < limit) limit = A.status->first_fail[1];
  if (limit > A.nq) limit = A.nq;
  // grouped order covers every query (rows at/after the first failure are skipped one by one)
  const uint64_t span = limit;
  const uint64_t ntiles = (span + tile_q - 1) / tile_q;
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t q0 = tile * tile_q;
    const uint32_t nq_here = (span - q0 < tile_q) ? (uint32_t)(span - q0) : tile_q;
    const uint32_t items = nq_here * LV;
    for (uint32_t it0 = threadIdx.x; it0 < items; it0 += TB * UNR) {
      bool live[UNR];
      uint32_t v[UNR], xi[UNR], yi[UNR];
      uint64_t qi[UNR];
      T x[UNR], y[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {          // phase 1: indices and query values
        const uint32_t it = it0 + (uint32_t)k * TB;
        live[k] = it < items;
        const uint32_t ql = live[k] ? it / LV : 0u;
        v[k] = live[k] ? it - ql * LV : 0u;
        qi[k] = q0 + ql;
        xi[k] = NDI_CHK(A.xi[qi[k]], (uint32_t)A.nx - 1u, BC_CELL_X);
        yi[k] = NDI_CHK(A.yi[qi[k]], (uint32_t)A.ny - 1u, BC_CELL_Y);
        x[k] = A.qx[qi[k]];
        y[k] = A.qy[qi[k]];
      }
      T x1[UNR], x2[UNR], y1[UNR], y2[UNR];
      V a11[UNR], a12[UNR], a21[UNR], a22[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {          // phase 2: knots and the four corner vectors
        x1[k] = xk[xi[k]]; x2[k] = xk[xi[k] + 1];
        y1[k] = yk[yi[k]]; y2[k] = yk[yi[k] + 1];
        const V* z11 = reinterpret_cast<const V*>(A.data + ((uint64_t)xi[k] * A.row_cells + yi[k]) * A.cell_elems);
        const V* z21 = reinterpret_cast<const V*>(A.data + ((uint64_t)(xi[k] + 1) * A.row_cells + yi[k]) * A.cell_elems);
        a11[k] = z11[v[k]];
        a12[k] = z11[LV + v[k]];                 // (xi,   yi+1)
        a21[k] = z21[v[k]];
        a22[k] = z21[LV + v[k]];                 // (xi+1, yi+1)
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) {          // phase 3: bilinear.rs:88-97, store
        V* o = reinterpret_cast<V*>(A.out + qi[k] * A.out_stride);
        if constexpr (SDIV) {   // the row's divisions share their divisors: one reciprocal per axis and item
          const SharedDivisor<T> dx = shared_divisor<T>(x2[k] - x1[k]), dy = shared_divisor<T>(y2[k] - y1[k]);
          const V z1 = frac_shared<T, V>(x1[k], a11[k], dx, a21[k], x[k]);
          const V z2 = frac_shared<T, V>(x1[k], a12[k], dx, a22[k], x[k]);
          if (live[k]) __builtin_nontemporal_store(frac_shared<T, V>(y1[k], z1, dy, z2, y[k]), o + v[k]);
          continue;
        }
        const V z1 = frac_v<T, V, APPROX>(x1[k], a11[k], x2[k], a21[k], x[k]);
        const V z2 = frac_v<T, V, APPROX>(x1[k], a12[k], x2[k], a22[k], x[k]);
        // written once, never re-read by the kernel: non-temporal (C3 -3 %, C5 share -4 %: profiles/r03_c3_grouped.md)
        if (live[k]) __builtin_nontemporal_store(frac_v<T, V, APPROX>(y1[k], z1, y2[k], z2, y[k]), o + v[k]);
      }
    }
  }
}

// TILE-GROUPED 2-D evaluation (ndi_path BUCKETED / AUTO for long batches on large grids).  The queries have been
// grouped by the tile of 2^ts x 2^ts cells their cell falls in (locate2_kernel's LDS histogram, group_offsets /
// bucket_scan, group_scatter2d_kernel), so the grouped positions [bin_start[b], bin_start[b+1]) all need corner rows
// from the same (2^ts + 1)^2 grid points.  A workgroup takes `chunk` consecutive grouped positions, and for every
// tile the chunk overlaps it copies the tile's grid points and knots into LDS ONCE (coalesced 16-byte loads of
// (2^ts + 1) contiguous row segments) and evaluates the tile's queries out of LDS: every grid value is read from
// memory once per (tile, chunk) instead of four times per query -- at C3 (2.4 queries per cell) 1.2 GB instead of
// 10.4 GB.  (Ordering the queries by tile alone and letting L2 do the re-use was measured first: the fabric still
// saw 5.2 GB -- concurrent misses on a line are not merged -- and the kernel stayed latency-bound at 1.45-1.65 ms;
// profiles/r03_c3_grouped.md.)  Results are bit-identical: same operands, same operation order (bilinear.rs:88-97).
// LV = lanes / VEC vectors per row must divide BLOCK.  XCD-aware chunk order as in eval_bucketed_kernel.
// TB threads per workgroup; records are handed over RB at a time (RB <= TB; the first RB threads load them); MAXI =
// 16-byte vectors per thread of the register double buffer ((2^ts + 1)^2 * LV <= MAXI * TB, checked by the host);
// COMPACT = self-contained 16-byte f32 records (no {qx, qy} side array, no LDS for it).  The host instantiates
// <1024, 1024, 6> (one workgroup per CU beside a tile of up to 96 KiB).  <512, 256, 10> -- two workgroups per CU, each
// with its own 74 KiB tile, the same 16 waves per CU but no shared barriers -- was measured at C3 and is SLOWER
// (1.54 vs 1.17 ms, profiles/r04_c3_tiles_ab.jsonl: twice the tile staging per CU and half the rows per trip); it
// stays selectable with NDI_TILE_WG=512 for A/B runs.
// SLOPE: the x-direction slopes m = (z[x+1][y] - z[x][y]) / (kx[x+1] - kx[x]) of every grid point of the tile are formed
// ONCE when the tile is staged (IEEE division, the reference's operands and operation: linear.rs:33) and kept in LDS
// next to the values: a query then needs ONE division per channel (the y direction, whose operands depend on the
// query) instead of three -- the same bits, because the slope of a grid cell does not depend on the query.  At C3
// (2.4 queries per cell) the tile kernel is bound by the arithmetic of those divisions.
template <class T, int VEC, int TB, int RB = TB, int MAXI = 6, bool COMPACT = false, bool SLOPE = false, int NREC = 1>
__global__ __launch_bounds__(TB, 4) void eval_bilinear_tiles_kernel(Eval2Args<T> A) {
  using V = typename VecT<T, VEC>::type;
  static_assert(RB <= TB, "the first RB threads of the workgroup load the records");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t BW = 64;                       // tiles whose start offsets are staged per chunk
  __shared__ uint32_t s_bs[BW + 1];
  __shared__ uint4 s_rec[RB];
  __shared__ T s_rq[COMPACT ? 2 : 2 * RB];   // {qx, qy} of non-compact records (f64, or an axis with more than 65536 knots)
  const uint32_t S = 1u << A.ts, S1 = S + 1u;
  // Channel split (A.ch_split = 2): the workgroup stages and evaluates only one half of the trailing axis -- LV is the
  // number of vectors per row IT handles, LVF the row's full length, `ch0` its first vector.  Two such workgroups of
  // 512 threads share a CU, each with its own half-tile (values + slopes: 72 KiB at C3), and overlap each other's
  // staging / slope / hand-over phases, which a single 1024-thread workgroup serialises behind its barriers.
  const uint32_t LVF = (uint32_t)(A.lanes / VEC);
  const uint32_t nsplit = A.ch_split ? A.ch_split : 1u;
  const uint32_t LV = LVF / nsplit;
  V* s_tile = reinterpret_cast<V*>(smem_raw);                              // [S1][S1][LV]
  V* s_mx = s_tile + (size_t)S1 * S1 * LV;                                 // SLOPE: [S][S1][LV]
  T* s_kx = reinterpret_cast<T*>(smem_raw + ((size_t)S1 * S1 + (SLOPE ? (size_t)S * S1 : 0)) * LV * sizeof(V));   // [S1]
  T* s_ky = s_kx + S1;
  T* s_rx = s_ky + S1;   // RN(1 / (kx[i+1] - kx[i])) per interval of the tile, 0 when the spacing is outside the
  T* s_dyr = s_rx + S1;  // divisor window (then the IEEE division is used); y direction: {ky[i+1] - ky[i], RN(1 / that)}
  const uint32_t tid = threadIdx.x;
  const uint32_t qpt = TB / LV;                 // queries per trip
  const uint32_t ql = tid / LV, v = tid - ql * LV;
  unsigned long long limit = A.status->first_fail[0];
  if (A.status->first_fail[1] < limit) limit = A.status->first_fail[1];
  if (limit > A.nq) limit = A.nq;
  const uint64_t n_pos = A.nq;                     // every query is grouped; rows at / after the first failure are skipped
  const uint64_t nchunks = (n_pos + A.chunk - 1) / A.chunk;
  const uint64_t per = (nchunks + 7) / 8;
  const bool xcd = (gridDim.x & 7u) == 0u;
  // (split: the workgroups of one chunk's halves are neighbours on the same XCD -- the second reader of the chunk's
  //  records finds them in that XCD's L2; the host makes the grid a multiple of 8 * ch_split)
  const uint32_t wg_local = xcd ? (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t ch0 = (wg_local % nsplit) * LV;
  const uint32_t lvs = 31u - (uint32_t)__builtin_clz(LV);   // LV is a power of two (it divides TB)
  const uint64_t c_first = (uint64_t)(wg_local / nsplit);
  const uint64_t c_step = (uint64_t)((xcd ? (gridDim.x >> 3) : gridDim.x) / nsplit);
  const uint64_t c_span = xcd ? per : nchunks;
  const uint64_t c_base = xcd ? (uint64_t)(blockIdx.x & 7u) * per : 0;
  for (uint64_t cl = c_first; cl < c_span; cl += c_step) {
    const uint64_t c = c_base + cl;
    if (c >= nchunks) break;
    const uint64_t p0 = c * A.chunk;
    const uint64_t p1 = (p0 + A.chunk < n_pos) ? p0 + A.chunk : n_pos;
    // The tile that holds position p0 comes from tile_chunk_bins_kernel (a 14-step binary search by one thread, with the
    // whole workgroup waiting, cost 6 % of the kernel at C3); the start offsets of the next BW tiles are staged in LDS
    // with one coalesced load, so that finding the next non-empty tile is LDS reads, not dependent global loads.
    __syncthreads();                                // (the previous chunk's window is no longer being read)
    const uint32_t b0 = A.chunk_bin[c];
    if (tid <= BW) {
      const uint32_t bb = b0 + tid;
      s_bs[tid] = (bb < A.nb) ? A.bin_start[bb] : (uint32_t)n_pos;   // (n_pos < 2^32: checked by the host)
    }
    __syncthreads();
    // Tiles are double-buffered through REGISTERS: while the records of the current tile are evaluated out of LDS,
    // the grid points, knots and knot spacings of the next tile the chunk touches are already on their way into
    // `pre` / `pk` (at most MAXI 16-byte vectors per thread); at the tile switch they are written to LDS.  vmcnt is
    // in-order, so the prefetch is complete by the time the first block of records of the current tile is handed
    // over -- its latency hides behind one block of evaluation instead of stalling the CU's only workgroup.
    // (2^ts + 1)^2 * LV <= MAXI * TB vectors: guaranteed by the host's tile budget
    struct TileGeo { uint64_t gx0, gy0, lo, hi; uint32_t rows, cols, b; bool valid; };
    auto find_tile = [&](uint32_t b_from) {
      TileGeo g{};
      g.valid = false;
      for (uint32_t bb = b_from; bb < A.nb; ++bb) {
        uint64_t bs, be;
        if (bb - b0 < BW) {                         // inside the staged window (entry BW is the window's upper neighbour)
          bs = s_bs[bb - b0];
          be = s_bs[bb - b0 + 1u];
        } else {
          bs = const_load(A.bin_start, bb);
          be = (bb + 1u < A.nb) ? (uint64_t)const_load(A.bin_start, bb + 1u) : n_pos;
        }
        if (bs >= p1) break;
        const uint64_t lo = bs > p0 ? bs : p0, hi = be < p1 ? be : p1;
        if (lo >= hi) continue;                     // empty tile (workgroup-uniform)
        const uint32_t tx = bb / A.nty, ty = bb - tx * A.nty;
        g.gx0 = (uint64_t)tx << A.ts;
        g.gy0 = (uint64_t)ty << A.ts;
        g.rows = (uint32_t)((A.nx - g.gx0 < S1) ? A.nx - g.gx0 : S1);   // grid points of the tile
        g.cols = (uint32_t)((A.ny - g.gy0 < S1) ? A.ny - g.gy0 : S1);
        g.lo = lo; g.hi = hi; g.b = bb; g.valid = true;
        break;
      }
      return g;
    };
    V pre[MAXI];
    uint32_t pre_off[MAXI];
    T pk0 = T(0), pk1 = T(0);
    auto prefetch = [&](const TileGeo& g) {
      const uint32_t row_vecs = g.cols * LV;        // one contiguous segment per grid row
      const uint32_t items = g.rows * row_vecs;
      const uint32_t rvm = g.cols == S1 ? A.rvm_full : A.rvm_edge;
      const T* tile0 = A.data + (g.gx0 * A.ny + g.gy0) * A.lanes;
      const uint32_t row_elems = (uint32_t)(A.ny * A.lanes);
      // Every thread issues exactly MAXI + 2 loads, whatever its share of the tile (indices are clamped, surplus values
      // are dropped at the commit): with the loads under conditions the compiler cannot count the operations behind
      // the record load that precedes them and waits for ALL of them -- the whole tile -- before the first records
      // are handed over (vmcnt is in-order).
#pragma unroll
      for (int k = 0; k < MAXI; ++k) {
        const uint32_t it = tid + (uint32_t)k * TB;
        const bool mine = it < items;
        const uint32_t itc = mine ? it : 0u;
        const uint32_t r = __umulhi(itc, rvm), j = itc - r * row_vecs;   // itc / row_vecs (itc * row_vecs < 2^32)
        // workgroup-uniform tile origin + one 32 x 32 -> 64-bit multiply-add per load (a grid row is < 2^32 elements:
        // checked by the host)
        const V* src = reinterpret_cast<const V*>(tile0 + (uint64_t)r * row_elems);
        pre[k] = src[(j >> lvs) * LVF + ch0 + (j & (LV - 1u))];   // (grid point, vector) of this workgroup's channel range
        pre_off[k] = mine ? r * S1 * LV + j : 0xffffffffu;
      }
      // threads 0 .. rows-1 carry the x knots (and their right neighbours, for the spacing), 128 .. 128+cols-1 the y
      // knots (a tile has at most 2^6 + 1 = 65 grid points per side)
      {
        const bool isy = tid >= 128u;
        const uint32_t t = isy ? tid - 128u : tid;
        const uint32_t len = isy ? g.cols : g.rows;
        const T* kk = isy ? A.yk + g.gy0 : A.xk + g.gx0;
        const uint32_t i0 = t < len ? t : len - 1u, i1 = t + 1u < len ? t + 1u : len - 1u;
        pk0 = kk[i0];
        pk1 = kk[i1];
      }
    };
    auto commit = [&](const TileGeo& g) {
#pragma unroll
      for (int k = 0; k < MAXI; ++k)
        if (pre_off[k] != 0xffffffffu) s_tile[pre_off[k]] = pre[k];
      // one IEEE reciprocal per knot interval of the tile (instead of one per division and lane)
      if (tid < g.rows) {
        s_kx[tid] = pk0;
        if (tid + 1u < g.rows) {
          const SharedDivisor<T> sd = shared_divisor<T>(pk1 - pk0);
          s_rx[tid] = sd.ok ? sd.r : T(0);
        }
      } else if (tid >= 128u && tid - 128u < g.cols) {
        s_ky[tid - 128u] = pk0;
        if (tid - 128u + 1u < g.cols) {
          const SharedDivisor<T> sd = shared_divisor<T>(pk1 - pk0);
          s_dyr[2u * (tid - 128u)] = sd.d;
          s_dyr[2u * (tid - 128u) + 1u] = sd.ok ? sd.r : T(0);
        }
      }
    };
    TileGeo cur = find_tile(b0);
    if (cur.valid) prefetch(cur);
    while (cur.valid) {
      const uint64_t gx0 = cur.gx0, gy0 = cur.gy0, lo = cur.lo, hi = cur.hi;
      __syncthreads();                              // the previous tile is no longer being read
      commit(cur);
      // the tile's first block of records is requested BEFORE the next tile: the wait for it then leaves the
      // MAXI + 2 loads of the prefetch outstanding (in-order vmcnt, static count) instead of draining them
      constexpr bool compact = COMPACT;
      const bool loader = (RB == TB) || tid < (uint32_t)RB;
      uint64_t pb = lo;
      uint4 r_in = make_uint4(0u, 0u, 0u, 0u);
      T rx_in = T(0), ry_in = T(0);
      {
        const uint64_t pc = (loader && pb + tid < hi) ? pb + tid : lo;   // clamped: one load per thread, always
        r_in = A.rec_i[pc];
        if (!compact) { rx_in = A.rec_q[2 * pc]; ry_in = A.rec_q[2 * pc + 1]; }
      }
      const TileGeo nxt = find_tile(cur.b + 1u);
      prefetch(nxt.valid ? nxt : cur);              // in flight while this tile's records are evaluated (after the
                                                    // chunk's last tile: the current one again, dropped -- never a branch)
      __syncthreads();
      if constexpr (SLOPE) {                        // the tile's values and knots are in LDS: form its x slopes
        const uint32_t row_vecs = cur.cols * LV;
        const uint32_t items = (cur.rows - 1u) * row_vecs;
        const uint32_t rvm = cur.cols == S1 ? A.rvm_full : A.rvm_edge;
        for (uint32_t it = tid; it < items; it += TB) {
          const uint32_t r = __umulhi(it, rvm), j = it - r * row_vecs;
          const uint32_t o = r * S1 * LV + j;
          // Linear::calc_frac's m (linear.rs:33) -- every vector of grid row r divides by the same knot spacing: the
          // correctly rounded shared-divisor division (IEEE division outside its window), as in the evaluation
          SharedDivisor<T> dx;
          dx.d = s_kx[r + 1u] - s_kx[r]; dx.r = s_rx[r]; dx.ok = dx.r > T(0);
          s_mx[o] = div_shared<T, V>(s_tile[o + S1 * LV] - s_tile[o], dx);
        }
      }
      // The tile's records are brought in RB at a time (one coalesced 16-byte load per thread) and handed to the
      // LV-lane groups through LDS; the next block's load is in flight while the current block is evaluated.
      // SLOPE: every record is decoded ONCE, by the thread that hands it over, instead of by each of the LV lanes that
      // evaluate it: {query index (all ones: at / after the batch's first failure), offset of the cell's corner in
      // the tile | y cell << 16, x - kx[cell], y - ky[cell]}.  (A thread past the end of the tile's records holds a
      // clamped / earlier record of this tile: decodable, never read.)
      auto hand_over = [&]() {
        if (!loader) return;
        if constexpr (SLOPE) {
          uint32_t xi, yi;
          T x, y;
          if (compact) {
            if constexpr (std::is_same<T, float>::value) {
              x = __builtin_bit_cast(float, r_in.z);
              y = __builtin_bit_cast(float, r_in.w);
            }
            xi = r_in.y & 0xffffu;
            yi = r_in.y >> 16;
          } else {
            x = rx_in;
            y = ry_in;
            xi = r_in.y;
            yi = r_in.z;
          }
          const uint64_t qi = NDI_CHK((uint64_t)r_in.x, A.nq, BC_QUERY);
          const uint32_t lx = NDI_CHK(xi - (uint32_t)gx0, S, BC_TILE), ly = NDI_CHK(yi - (uint32_t)gy0, S, BC_TILE);
          const uint32_t zo = (lx * S1 + ly) * LV;            // < MAXI * TB <= 2^16
          const T ddx = x - s_kx[lx], yd = y - s_ky[ly];      // linear.rs:35's (x - x1) of both directions
          const uint32_t qw = qi < limit ? (uint32_t)qi : 0xffffffffu;
          if (compact) {
            if constexpr (std::is_same<T, float>::value)
              s_rec[tid] = make_uint4(qw, zo | (ly << 16), __builtin_bit_cast(uint32_t, ddx), __builtin_bit_cast(uint32_t, yd));
          } else {
            s_rec[tid] = make_uint4(qw, zo | (ly << 16), 0u, 0u);
            s_rq[2 * tid] = ddx;
            s_rq[2 * tid + 1] = yd;
          }
        } else {
          s_rec[tid] = r_in;
          if (!compact) { s_rq[2 * tid] = rx_in; s_rq[2 * tid + 1] = ry_in; }
        }
      };
      auto request = [&](uint64_t p) {               // the block of records starting at grouped position p
        if (loader && p + tid < hi) {
          r_in = A.rec_i[p + tid];
          if (!compact) { rx_in = A.rec_q[2 * (p + tid)]; ry_in = A.rec_q[2 * (p + tid) + 1]; }
        }
      };
      auto evaluate = [&](uint32_t cnt) {
        if constexpr (SLOPE) {
          for (uint32_t j = ql; j < cnt; j += qpt) {
            const uint4 r = s_rec[j];
            if (r.x == 0xffffffffu) continue;
            const uint32_t ly = r.y >> 16, zo = (r.y & 0xffffu) + v;
            T ddx, yd;
            if (compact) {
              if constexpr (std::is_same<T, float>::value) {
                ddx = __builtin_bit_cast(float, r.z);
                yd = __builtin_bit_cast(float, r.w);
              }
            } else {
              ddx = s_rq[2 * j];
              yd = s_rq[2 * j + 1];
            }
            const V m1 = s_mx[zo], m2 = s_mx[zo + LV], b1 = s_tile[zo], b2 = s_tile[zo + LV];
            SharedDivisor<T> dy;
            dy.d = s_dyr[2u * ly]; dy.r = s_dyr[2u * ly + 1u]; dy.ok = dy.r > T(0);
            const V z1 = m1 * ddx + b1;   // m * (x - x1) + b with the staged slopes (bilinear.rs:88-97, linear.rs:33-35)
            const V z2 = m2 * ddx + b2;
            const V m = div_shared<T, V>(z2 - z1, dy);
            V* o = reinterpret_cast<V*>(A.out + (uint64_t)r.x * (uint32_t)A.out_stride) + ch0 + v;   // (row stride < 2^32: host)
#ifdef NDI_TUNING
            if (A.debug & 4) {
              const V w = m * yd + z1;
              if (w[0] == T(-123.456)) __builtin_nontemporal_store(w, o);
              continue;
            }
#endif
            __builtin_nontemporal_store(m * yd + z1, o);
          }
        } else {
          for (uint32_t j = ql; j < cnt; j += qpt) {
            const uint4 r = s_rec[j];
            uint32_t xi, yi;
            T x, y;
            if (compact) {
              if constexpr (std::is_same<T, float>::value) {
                x = __builtin_bit_cast(float, r.z);
                y = __builtin_bit_cast(float, r.w);
              }
              xi = r.y & 0xffffu;
              yi = r.y >> 16;
            } else {
              x = s_rq[2 * j];
              y = s_rq[2 * j + 1];
              xi = r.y;
              yi = r.z;
            }
            const uint64_t qi = NDI_CHK((uint64_t)r.x, A.nq, BC_QUERY);
            if (qi >= limit) continue;
            const uint32_t lx = NDI_CHK(xi - (uint32_t)gx0, S, BC_TILE), ly = NDI_CHK(yi - (uint32_t)gy0, S, BC_TILE);
            const uint32_t zo = (lx * S1 + ly) * LV + v;
            const V* z11 = s_tile + zo;
            const T y1 = s_ky[ly], y2 = s_ky[ly + 1u];
            V* o = reinterpret_cast<V*>(A.out + qi * A.out_stride) + ch0;
            SharedDivisor<T> dy;
            dy.d = y2 - y1; dy.r = s_dyr[2u * ly + 1u]; dy.ok = dy.r > T(0);
            const V a11 = z11[0], a12 = z11[LV], a21 = z11[(size_t)S1 * LV], a22 = z11[(size_t)S1 * LV + LV];
            const T x1 = s_kx[lx], x2 = s_kx[lx + 1u];
            SharedDivisor<T> dx;
            dx.d = x2 - x1; dx.r = s_rx[lx]; dx.ok = dx.r > T(0);
            const V z1 = frac_shared<T, V>(x1, a11, dx, a21, x);   // bilinear.rs:88-97
            const V z2 = frac_shared<T, V>(x1, a12, dx, a22, x);
            __builtin_nontemporal_store(frac_shared<T, V>(y1, z1, dy, z2, y), o + v);
          }
        }
      };
      // The first block is handed over HERE, outside the loop: the only memory operations between its load and this
      // use are the MAXI + 2 unconditional loads of the prefetch, so the compiler waits with an exact count
      // (vmcnt(MAXI + 2): the records only) and the next tile stays in flight during the whole evaluation below.
      // (Inside the loop the same use follows a conditional load and waits for everything -- which, for the first
      // block of every tile, meant: for the whole next tile, before a single record was evaluated.)
      uint32_t cnt = (hi - pb < (uint64_t)RB) ? (uint32_t)(hi - pb) : (uint32_t)RB;
      hand_over();
      uint64_t nxt_p = pb + RB;
      request(nxt_p);
      __syncthreads();                              // records (and slopes) of the tile are in LDS
      for (;;) {
        evaluate(cnt);
        pb = nxt_p;
        if (pb >= hi) break;
        cnt = (hi - pb < (uint64_t)RB) ? (uint32_t)(hi - pb) : (uint32_t)RB;
        __syncthreads();                            // the previous block's records are no longer being read
        hand_over();
        nxt_p = pb + RB;
        request(nxt_p);
        __syncthreads();
      }
      cur = nxt;
    }
  }
}

// 2-D GATHER order for rows of fewer than 256 vectors on axes that fit LDS twice over, QUERY ORDER with both searches
// fused in -- the 2-D analogue of eval_fused_kernel, for the reference's own 2-D shapes (100 x 100 x 5,
// benches/bench_interp2d.rs) and few-channel grids.  A wave takes 64 consecutive queries, one per lane: both searches
// (bucket index or pyramid, staged in LDS), the cell's offset in the grid, (x - x1), (y - y1) and the two knot
// spacings with their correctly rounded reciprocals (ONE IEEE division per direction and QUERY instead of three
// divisions per value), parked in a wave-private LDS strip; then the wave walks the batch's 64 * LV output vectors in
// row-major order, 64 per trip: four corner loads, three shared-divisor divisions (div_shared: the bits of the IEEE
// divisions of bilinear.rs:88-97), one store -- a trip's stores are 1 KiB of consecutive output bytes.  No (xi, yi)
// round trip through memory, no second launch.  The first failing query is known before the launch
// (range_check_kernel): rows at / after it are never written (interp2d/mod.rs:297-306).
template <class T>
struct EvalFused2Args {
  Pyramid<T> px, py;
  BucketIndex<T> bx, by;   // lut == nullptr: pyramid search on that axis
  const T* data;           // plain or pair-packed grid (pack_pairs_kernel)
  const T* qx;
  const T* qy;
  T* out;
  uint64_t nq, out_stride;
  uint32_t lanes;
  uint32_t lv;             // vectors per row
  uint32_t lv_magic;       // ceil(2^32 / lv) for lv >= 2
  uint32_t cell_vecs;      // vectors between z[xi][yi] and z[xi][yi+1]'s slot: lv (plain grid) or 2 lv (pair-packed)
  uint32_t row_vecs;       // vectors between grid rows xi and xi + 1
  int mode;
  unsigned long long* first_fail;   // [2]: x, y (range_check_kernel, or this kernel when `check`)
  int check;                        // see EvalFusedArgs
};

template <class T>
__device__ __forceinline__ void lane_check2(unsigned long long* first_fail, uint64_t qi, T x, T y, T x0, T xn, T y0, T yn,
                                            int mode);

template <class T, int VEC, int UNR, int TB>
__global__ __launch_bounds__(TB) void eval_fused2d_kernel(EvalFused2Args<T> A) {
  using V = typename VecT<T, VEC>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t WAVES = TB / 64;
  if (A.nq == 0) return;
  const uint32_t tid = threadIdx.x;
  const uint32_t nxa = A.px.n + A.px.n1, nya = A.py.n + A.py.n1;
  // LDS: [x pyramid | y pyramid | x lut | y lut | per-wave strips: cell offset, fx, fy, dx, rx, dy, ry]
  size_t off;
  {
    T* sx = reinterpret_cast<T*>(smem_raw);
    T* sy = sx + nxa;
    for (uint32_t i = tid; i < nxa; i += TB) sx[i] = A.px.lv0[i];
    for (uint32_t i = tid; i < nya; i += TB) sy[i] = A.py.lv0[i];
    off = ((size_t)(nxa + nya) * sizeof(T) + 15u) & ~(size_t)15u;
  }
  lds_u16 lutx = nullptr, luty = nullptr;
  if (A.bx.lut || A.by.lut) {
    uint32_t* sl = reinterpret_cast<uint32_t*>(smem_raw + off);
    const uint32_t wx = A.bx.lut ? (A.bx.m + 2u) / 2u : 0u, wy = A.by.lut ? (A.by.m + 2u) / 2u : 0u;
    const uint32_t* srcx = reinterpret_cast<const uint32_t*>(A.bx.lut);
    const uint32_t* srcy = reinterpret_cast<const uint32_t*>(A.by.lut);
    for (uint32_t i = tid; i < wx; i += TB) sl[i] = srcx[i];
    for (uint32_t i = tid; i < wy; i += TB) sl[wx + i] = srcy[i];
    if (wx) lutx = (lds_u16)(smem_raw + off);
    if (wy) luty = (lds_u16)(smem_raw + off + (size_t)wx * 4u);
    off += (((size_t)(wx + wy) * 4u) + 15u) & ~(size_t)15u;
  }
  uint32_t* w_o = reinterpret_cast<uint32_t*>(smem_raw + off) + (tid >> 6) * 64u;
  off += (size_t)WAVES * 64u * sizeof(uint32_t);
  T* w_s = reinterpret_cast<T*>(smem_raw + off) + (tid >> 6) * 64u * 6u;   // [6][64] per wave: fx, fy, dx, rx, dy, ry
  __syncthreads();
  PyramidLds<T> PX, PY;
  PX.lv0 = (lds_ptr<T>)(smem_raw);
  PX.lv1 = PX.lv0 + A.px.n;
  PX.n = A.px.n; PX.n1 = A.px.n1; PX.levels = A.px.levels; PX.guess = A.px.guess; PX.block = A.px.block;
  PY.lv0 = PX.lv0 + nxa;
  PY.lv1 = PY.lv0 + A.py.n;
  PY.n = A.py.n; PY.n1 = A.py.n1; PY.levels = A.py.levels; PY.guess = A.py.guess; PY.block = A.py.block;
  const T x0 = PX.lv0[0], xn = PX.lv0[PX.n - 1], y0 = PY.lv0[0], yn = PY.lv0[PY.n - 1];
  const uint32_t lane = tid & 63u;
  const uint32_t LV = A.lv;
  const bool contig = A.out_stride == (uint64_t)A.lanes;
  unsigned long long limit = A.check ? NO_FAIL : (A.first_fail[0] < A.first_fail[1] ? A.first_fail[0] : A.first_fail[1]);
  if (limit > A.nq) limit = A.nq;
  const V* const G = reinterpret_cast<const V*>(A.data);
  const uint64_t wave_step = (uint64_t)gridDim.x * TB;
  // queries of QB batches requested together, one round ahead (see eval_fused_kernel: one store drain per QB batches)
  constexpr int QB = 4;
  uint64_t round0 = ((uint64_t)blockIdx.x * WAVES + (tid >> 6)) * 64u;
  T xq[QB], yq[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const uint64_t p = round0 + (uint64_t)j * wave_step + lane;
    const uint64_t pc = p < A.nq ? p : A.nq - 1u;
    xq[j] = A.qx[pc];
    yq[j] = A.qy[pc];
  }
  for (; round0 < limit; round0 += (uint64_t)QB * wave_step) {
  T xc[QB], yc[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const bool in = round0 + (uint64_t)j * wave_step + lane < limit;
    xc[j] = in ? xq[j] : x0;
    yc[j] = in ? yq[j] : y0;
  }
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const uint64_t p = round0 + (uint64_t)(QB + j) * wave_step + lane;
    const uint64_t pc = p < A.nq ? p : A.nq - 1u;
    xq[j] = A.qx[pc];
    yq[j] = A.qy[pc];
  }
#pragma unroll 1
  for (int jb = 0; jb < QB; ++jb) {
    const uint64_t base = round0 + (uint64_t)jb * wave_step;
    if (base >= limit) break;
    T x = xc[0], y = yc[0];
#pragma unroll
    for (int j = 1; j < QB; ++j)
      if (jb == j) { x = xc[j]; y = yc[j]; }
    if (A.check && base + lane < limit) lane_check2<T>(A.first_fail, base + lane, x, y, x0, xn, y0, yn, A.mode);   // fresh output
    const uint32_t xi = lutx ? locate_index_lut<T>(PX, lutx, A.bx.m, A.bx.scale, x0, xn, x)
                             : locate_index<T, lds_ptr<T>>(PX, x0, xn, x, lane);   // all 64 lanes take part
    const uint32_t yi = luty ? locate_index_lut<T>(PY, luty, A.by.m, A.by.scale, y0, yn, y)
                             : locate_index<T, lds_ptr<T>>(PY, y0, yn, y, lane);
    {
      const T x1 = PX.lv0[xi], x2 = PX.lv0[xi + 1], y1 = PY.lv0[yi], y2 = PY.lv0[yi + 1];
      const SharedDivisor<T> dx = shared_divisor<T>(x2 - x1), dy = shared_divisor<T>(y2 - y1);
      w_o[lane] = NDI_CHK(xi, PX.n - 1u, BC_CELL_X) * A.row_vecs + NDI_CHK(yi, PY.n - 1u, BC_CELL_Y) * A.cell_vecs;
      w_s[0 * 64 + lane] = x - x1;              // linear.rs:35's (x - x1) of both directions
      w_s[1 * 64 + lane] = y - y1;
      w_s[2 * 64 + lane] = dx.d;
      w_s[3 * 64 + lane] = dx.ok ? dx.r : T(0);
      w_s[4 * 64 + lane] = dy.d;
      w_s[5 * 64 + lane] = dy.ok ? dy.r : T(0);
    }
    __builtin_amdgcn_wave_barrier();        // LDS operations of one wave execute in order: no s_barrier needed
    const uint32_t nq_here = (limit - base < 64u) ? (uint32_t)(limit - base) : 64u;
    const uint32_t items = nq_here * LV;
    T* const o_base = A.out + base * A.out_stride;
    for (uint32_t it0 = lane; it0 < items; it0 += 64u * UNR) {
      bool live[UNR];
      uint32_t ql[UNR], v[UNR], e[UNR];
      T fx[UNR], fy[UNR];
      SharedDivisor<T> dx[UNR], dy[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {        // phase 1: (query, vector) of the item and the query's scalars
        const uint32_t it = it0 + (uint32_t)k * 64u;
        live[k] = it < items;
        const uint32_t itc = live[k] ? it : 0u;
        ql[k] = (LV == 1u) ? itc : __umulhi(itc, A.lv_magic);
        ql[k] = NDI_CHK(ql[k], 64u, BC_STRIP);
        v[k] = itc - ql[k] * LV;
        e[k] = w_o[ql[k]] + v[k];
        fx[k] = w_s[0 * 64 + ql[k]];
        fy[k] = w_s[1 * 64 + ql[k]];
        dx[k].d = w_s[2 * 64 + ql[k]]; dx[k].r = w_s[3 * 64 + ql[k]]; dx[k].ok = dx[k].r > T(0);
        dy[k].d = w_s[4 * 64 + ql[k]]; dy[k].r = w_s[5 * 64 + ql[k]]; dy[k].ok = dy[k].r > T(0);
      }
      V a11[UNR], a12[UNR], a21[UNR], a22[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {        // phase 2: the four corner vectors
        a11[k] = G[e[k]];
        a12[k] = G[e[k] + LV];
        a21[k] = G[e[k] + A.row_vecs];
        a22[k] = G[e[k] + A.row_vecs + LV];
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) {        // phase 3: bilinear.rs:88-97 and the store
        const V z1 = div_shared<T, V>(a21[k] - a11[k], dx[k]) * fx[k] + a11[k];
        const V z2 = div_shared<T, V>(a22[k] - a12[k], dx[k]) * fx[k] + a12[k];
        const V r = div_shared<T, V>(z2 - z1, dy[k]) * fy[k] + z1;
        V* o = contig ? reinterpret_cast<V*>(o_base) + (it0 + (uint32_t)k * 64u)
                      : reinterpret_cast<V*>(o_base + (uint64_t)ql[k] * A.out_stride) + v[k];
        if (live[k]) store_stream<true>(o, r);
      }
    }
    __builtin_amdgcn_wave_barrier();        // the strip is rewritten by the next batch
  }
  }
}

// 2-D QUERY PER LANE with the whole GRID resident in LDS -- the reference's own 2-D bench shape (a 100 x 100 scalar grid,
// benches/bench_interp2d.rs:12-18: 80 KB in f64) and any other grid that fits beside its axes.  The counters of the
// query-order kernel on that shape (profiles/r05_small_shapes_counters.txt) show the L1 address path as the bound: every
// one of the four corner loads of a wave touches up to 64 different cache lines (374 L1 accesses per 64 queries: 0.48 of
// the 0.90 ms).  With the grid in LDS no corner read leaves the CU.  A lane owns its query: both searches branch-free
// through the axes' dense bucket indices (lane_axis_index), the two knot spacings with their staged reciprocals (one record
// {k_l, dk, RN(1 / dk)} per knot interval and axis), the four corners of every value from LDS, bilinear.rs:88-97 with the
// correctly rounded shared-divisor divisions (div_shared: the bits of the IEEE divisions).  eval_scalar2d_kernel (one value
// per grid point): QPL consecutive queries per lane, 16-byte query loads and stores; eval_lanes2d_kernel (rows of several
// values): out through a wave-private strip as one sequential stream of 16-byte vectors.
template <class T>
struct EvalLanes2Args {
  const T* xk;             // [nx]
  const T* yk;             // [ny]
  uint32_t nx, ny;
  DenseLut<T> dx, dy;
  const T* data;           // plain grid [nx][ny][lanes]
  const T* qx;
  const T* qy;
  T* out;
  uint64_t nq, out_stride;
  uint32_t lanes;
  int mode;
  unsigned long long* first_fail;   // [2]: x, y (range_check_kernel, or this kernel when `check`)
  int check;                        // see EvalLanesArgs
};

// the failure conditions of a 2-D query (Interp2D::is_in_x_range / is_in_y_range, interp2d/mod.rs:374-379): x and y
// failures are recorded separately, the host reports x before y for the same query (bilinear.rs:71-80)
template <class T>
__device__ __forceinline__ void lane_check2(unsigned long long* first_fail, uint64_t qi, T x, T y, T x0, T xn, T y0, T yn,
                                            int mode) {
  const bool badx = (mode == EX_NO) ? !((x0 <= x) && (x <= xn)) : !(x == x);
  const bool bady = (mode == EX_NO) ? !((y0 <= y) && (y <= yn)) : !(y == y);
  if (badx) atomicMin(&first_fail[0], (unsigned long long)qi);
  if (bady) atomicMin(&first_fail[1], (unsigned long long)qi);
}

template <class T>
struct LaneCell {
  uint32_t o;              // element offset of the cell's first corner
  T fx, fy;
  SharedDivisor<T> dx, dy;
};

template <class T>
__device__ __forceinline__ LaneCell<T> lane_cell(const LaneAxis<T>& SX, const LaneAxis<T>& SY, const XRecs<T>& XX,
                                                 const XRecs<T>& XY, uint32_t ny, uint32_t L, T x, T y) {
  const uint32_t xi = NDI_CHK(lane_axis_index<T>(SX, x), SX.n - 1u, BC_CELL_X);
  const uint32_t yi = NDI_CHK(lane_axis_index<T>(SY, y), SY.n - 1u, BC_CELL_Y);
  T rx[4], ry[4];
  XX.get(xi, rx);
  XY.get(yi, ry);
  LaneCell<T> c;
  c.o = (xi * ny + yi) * L;
  c.fx = x - rx[0];                   // linear.rs:35's (x - x1) of both directions
  c.fy = y - ry[0];
  c.dx.d = rx[1]; c.dx.r = rx[2]; c.dx.ok = rx[2] > T(0);
  c.dy.d = ry[1]; c.dy.r = ry[2]; c.dy.ok = ry[2] > T(0);
  return c;
}

template <class T>
__device__ __forceinline__ T lane_bilinear(const T* g, uint32_t L, uint32_t rowe, const LaneCell<T>& c) {   // bilinear.rs:88-97
  const T a11 = g[0], a12 = g[L], a21 = g[rowe], a22 = g[rowe + L];
  const T z1 = div_shared<T, T>(a21 - a11, c.dx) * c.fx + a11;
  const T z2 = div_shared<T, T>(a22 - a12, c.dx) * c.fx + a12;
  return div_shared<T, T>(z2 - z1, c.dy) * c.fy + z1;
}

// stages both axes, their interval records and the grid; returns the LDS offset behind them
template <class T, int TB>
__device__ __forceinline__ size_t stage_grid2(unsigned char* smem, const EvalLanes2Args<T>& A, LaneAxis<T>& SX, LaneAxis<T>& SY,
                                              XRecs<T>& XX, XRecs<T>& XY, T*& s_g) {
  constexpr int VN = Wide<T>::N;
  using V = typename VecT<T, VN>::type;
  size_t off = 0;
  SX = stage_lane_axis<T, TB>(smem, off, A.xk, A.nx, A.dx);
  SY = stage_lane_axis<T, TB>(smem, off, A.yk, A.ny, A.dy);
  XX = XRecs<T>{reinterpret_cast<typename XRecs<T>::U*>(smem + off), A.nx - 1u};
  off += XRecs<T>::bytes(A.nx - 1u);
  XY = XRecs<T>{reinterpret_cast<typename XRecs<T>::U*>(smem + off), A.ny - 1u};
  off += XRecs<T>::bytes(A.ny - 1u);
  stage_xrecs<T, TB>(XX, A.xk, A.nx);
  stage_xrecs<T, TB>(XY, A.yk, A.ny);
  s_g = reinterpret_cast<T*>(smem + off);
  const uint32_t gelems = A.nx * A.ny * A.lanes;
  if ((gelems % VN) == 0u && (reinterpret_cast<uintptr_t>(A.data) & 15u) == 0u) {
    const V* src = reinterpret_cast<const V*>(A.data);
    V* dst = reinterpret_cast<V*>(s_g);
    for (uint32_t i = threadIdx.x; i < gelems / VN; i += TB) dst[i] = src[i];
  } else {
    for (uint32_t i = threadIdx.x; i < gelems; i += TB) s_g[i] = A.data[i];
  }
  off += ((size_t)gelems * sizeof(T) + 15u) & ~(size_t)15u;
  return off;
}

template <class T, int QPL, int TB>
__global__ __launch_bounds__(TB) void eval_scalar2d_kernel(EvalLanes2Args<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using QV = typename VecT<T, QPL>::type;
  if (A.nq == 0) return;
  const uint32_t tid = threadIdx.x;
  LaneAxis<T> SX, SY;
  XRecs<T> XX, XY;
  T* s_g;
  stage_grid2<T, TB>(smem_raw, A, SX, SY, XX, XY, s_g);
  __syncthreads();
  unsigned long long limit = A.check ? NO_FAIL : (A.first_fail[0] < A.first_fail[1] ? A.first_fail[0] : A.first_fail[1]);
  if (limit > A.nq) limit = A.nq;
  const uint32_t ny = A.ny;
  const uint64_t nvec = limit / QPL;
  const uint64_t step = (uint64_t)gridDim.x * TB;
  const QV* qxv = reinterpret_cast<const QV*>(A.qx);
  const QV* qyv = reinterpret_cast<const QV*>(A.qy);
  QV* ov = reinterpret_cast<QV*>(A.out);
  const uint64_t vlast = nvec ? nvec - 1u : 0u;
  auto eval_vec = [&](uint64_t v, const QV& cx, const QV& cy) -> QV {
    QV res;
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
      T x, y;
      if constexpr (QPL == 1) { x = cx; y = cy; } else { x = cx[u]; y = cy[u]; }
      if (A.check) lane_check2<T>(A.first_fail, v * QPL + u, x, y, SX.k0, SX.kn, SY.k0, SY.kn, A.mode);
      const LaneCell<T> c = lane_cell<T>(SX, SY, XX, XY, ny, 1u, x, y);
      const T r = lane_bilinear<T>(s_g + c.o, 1u, ny, c);
      if constexpr (QPL == 1) res = r; else res[u] = r;
    }
    return res;
  };
  auto storeq = [&](uint64_t v, const QV& res) {
    if constexpr (QPL == 1) store_stream<true>(A.out + v * A.out_stride, res);
    else store_stream<true>(ov + v, res);
  };
  // two vectors of (x, y) queries per thread and trip, the next two in flight (see eval_scalar_kernel)
  uint64_t vi = (uint64_t)blockIdx.x * TB + tid;
  QV nx0 = QV(SX.k0), ny0 = QV(SY.k0), nx1 = QV(SX.k0), ny1 = QV(SY.k0);
  if (nvec) {
    const uint64_t v0 = vi < nvec ? vi : vlast, v1 = vi + step < nvec ? vi + step : vlast;
    nx0 = qxv[v0]; ny0 = qyv[v0];
    nx1 = qxv[v1]; ny1 = qyv[v1];
  }
  for (; vi < nvec; vi += 2u * step) {
    const QV cx0 = nx0, cy0 = ny0, cx1 = nx1, cy1 = ny1;
    {
      const uint64_t a = vi + 2u * step, b = vi + 3u * step;
      const uint64_t v0 = a < nvec ? a : vlast, v1 = b < nvec ? b : vlast;
      nx0 = qxv[v0]; ny0 = qyv[v0];
      nx1 = qxv[v1]; ny1 = qyv[v1];
    }
    storeq(vi, eval_vec(vi, cx0, cy0));
    if (vi + step < nvec) storeq(vi + step, eval_vec(vi + step, cx1, cy1));
  }
  if constexpr (QPL > 1) {
    const uint64_t qi = nvec * QPL + tid;
    if (blockIdx.x == 0 && qi < limit) {
      if (A.check) lane_check2<T>(A.first_fail, qi, A.qx[qi], A.qy[qi], SX.k0, SX.kn, SY.k0, SY.kn, A.mode);
      const LaneCell<T> c = lane_cell<T>(SX, SY, XX, XY, ny, 1u, A.qx[qi], A.qy[qi]);
      A.out[qi * A.out_stride] = lane_bilinear<T>(s_g + c.o, 1u, ny, c);
    }
  }
}

template <class T, int TB>
__global__ __launch_bounds__(TB) void eval_lanes2d_kernel(EvalLanes2Args<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int VN = Wide<T>::N;
  using V = typename VecT<T, VN>::type;
  if (A.nq == 0) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, L = A.lanes;
  LaneAxis<T> SX, SY;
  XRecs<T> XX, XY;
  T* s_g;
  const size_t off = stage_grid2<T, TB>(smem_raw, A, SX, SY, XX, XY, s_g);
  T* s_strip = reinterpret_cast<T*>(smem_raw + off) + (size_t)(tid >> 6) * 64u * L;
  __syncthreads();
  unsigned long long limit = A.check ? NO_FAIL : (A.first_fail[0] < A.first_fail[1] ? A.first_fail[0] : A.first_fail[1]);
  if (limit > A.nq) limit = A.nq;
  const uint32_t ny = A.ny, rowe = ny * L;
  // (16-byte vector stores of the batch's rows: the rows must be contiguous AND the buffer 16-byte aligned -- a sliced
  //  view such as out[1:] with 5 lanes takes the per-element path)
  const bool contig = A.out_stride == (uint64_t)L && (reinterpret_cast<uintptr_t>(A.out) & 15u) == 0u;
  const uint32_t rot = (L & 1u) ? 0u : lane % L;
  const uint64_t wstep = (uint64_t)gridDim.x * TB;
  uint64_t base = ((uint64_t)blockIdx.x * (TB / 64) + (tid >> 6)) * 64u;
  T xq, yq;
  {
    const uint64_t pc = (base + lane < A.nq) ? base + lane : A.nq - 1u;
    xq = A.qx[pc];
    yq = A.qy[pc];
  }
  for (; base < limit; base += wstep) {
    const T x = xq, y = yq;
    {
      const uint64_t pn = base + wstep + lane;
      const uint64_t pc = pn < A.nq ? pn : A.nq - 1u;
      xq = A.qx[pc];
      yq = A.qy[pc];
    }
    if (A.check && base + lane < limit) lane_check2<T>(A.first_fail, base + lane, x, y, SX.k0, SX.kn, SY.k0, SY.kn, A.mode);
    const LaneCell<T> c = lane_cell<T>(SX, SY, XX, XY, ny, L, x, y);
    T* mine = s_strip + lane * L;
    for (uint32_t k = 0; k < L; ++k) {      // (even L: rotated start, see eval_lanes_kernel)
      uint32_t l = k + rot;
      if (l >= L) l -= L;
      mine[l] = lane_bilinear<T>(s_g + c.o + l, L, rowe, c);
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t nq_here = (limit - base < 64u) ? (uint32_t)(limit - base) : 64u;
    const uint32_t total = nq_here * L;
    if (contig) {
      T* const o = A.out + base * L;
      for (uint32_t e0 = lane * VN; e0 < total; e0 += 64u * VN) {
        if (e0 + VN <= total) {
          store_stream<true>(reinterpret_cast<V*>(o + e0), *reinterpret_cast<const V*>(s_strip + e0));
        } else {
          for (uint32_t e = e0; e < total; ++e) o[e] = s_strip[e];
        }
      }
    } else {
      for (uint32_t it = lane; it < total; it += 64u) {
        const uint32_t ql = it / L, l = it - ql * L;
        A.out[(base + ql) * A.out_stride + l] = s_strip[it];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// SLOPE RECORDS for short rows on grids that do not fit LDS (round 6; the reference's 100 x 100 x 5 bench grid,
// benches/bench_interp2d.rs:87-92).  Bilinear::interp_into (bilinear.rs:88-97) forms, per channel,
//     z1 = calc_frac((x1, z11), (x2, z21), x) = m1 (x - x1) + z11,   m1 = (z21 - z11) / (x2 - x1)      (linear.rs:33-35)
//     z2 = calc_frac((x1, z12), (x2, z22), x) = m2 (x - x1) + z12,   m2 = (z22 - z12) / (x2 - x1)
//     z  = calc_frac((y1, z1), (y2, z2), y).
// m1 and m2 have no query in them: they are formed ONCE per grid point by the IEEE division of the same operands (the same
// bits) and kept next to the value, SR[xi][yi][c] = { z[xi][yi][c], m[xi][yi][c] } for xi < nx - 1 (slope_pack_kernel; 2 x
// the grid, built on first use).  A query then needs ONE contiguous run of 4 L values -- the records of (xi, yi) and
// (xi, yi + 1) -- two of its three divisions per channel are gone, and nothing has to cross lanes: a lane owns one
// (query, channel) item, loads its two 2-value records (16 bytes each in f64) next to its neighbours' and evaluates the
// whole bilinear form.  The 64 L items of a wave's 64 queries are walked in row-major order, 64 per trip, exactly L
// full trips, so every trip's loads of one query are neighbouring lanes on neighbouring addresses (about 2-3 L1 accesses
// per query instead of 11.9) and every trip's stores are 64 consecutive output values: one sequential write stream.
// The per-query scalars {cell offset, x - x1, y - y1, dy, RN(1 / dy)} are parked in a wave-private LDS strip as
// 16-byte units (three reads per trip, no bank conflicts: neighbouring queries, neighbouring units).
template <class T>
__global__ __launch_bounds__(BLOCK) void slope_pack_kernel(const T* data, const T* xk, T* out, uint64_t nx, uint64_t ny,
                                                           uint64_t lanes, uint64_t row_cells, uint64_t cell_elems,
                                                           uint64_t cell_stride) {
  // element (xi, yi, c) of the source: data[(xi * row_cells + yi) * cell_elems + c] in either layout (pair-packed,
  // pack_pairs_kernel: the last grid column is reached as the second half of pair ny - 2)
  auto src = [&](uint64_t xi, uint64_t yi, uint64_t c) -> T {
    if (cell_elems == lanes) return data[(xi * row_cells + yi) * cell_elems + c];
    return yi < row_cells ? data[(xi * row_cells + yi) * cell_elems + c] : data[(xi * row_cells + yi - 1) * cell_elems + lanes + c];
  };
  if (cell_stride == 0) {       // POINT records: SR[xi][yi][c] = {z, m}, a query's two records are neighbours
    const uint64_t per_row = ny * lanes, total = (nx - 1) * per_row;
    for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (uint64_t)gridDim.x * BLOCK) {
      const uint64_t xi = e / per_row, r = e - xi * per_row;
      const uint64_t yi = r / lanes, c = r - yi * lanes;
      const T z = src(xi, yi, c), zn = src(xi + 1, yi, c);
      out[2 * e] = z;
      out[2 * e + 1] = (zn - z) / (xk[xi + 1] - xk[xi]);       // linear.rs:33, once per grid point
    }
    return;
  }
  // CELL records: every cell (xi, yi) owns {z, m}[yi][0 .. L) ++ {z, m}[yi + 1][0 .. L) at (xi (ny - 1) + yi) cell_stride
  // elements -- the stride padded so that a record never straddles more 128-byte lines than it has to
  const uint64_t per_cell = 2 * lanes, per_row = (ny - 1) * per_cell, total = (nx - 1) * per_row;
  for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (uint64_t)gridDim.x * BLOCK) {
    const uint64_t xi = e / per_row, r = e - xi * per_row;
    const uint64_t yc = r / per_cell, k = r - yc * per_cell;
    const uint64_t yi = yc + (k >= lanes ? 1 : 0), c = k >= lanes ? k - lanes : k;
    const T z = src(xi, yi, c), zn = src(xi + 1, yi, c);
    T* o = out + (xi * (ny - 1) + yc) * cell_stride + 2 * k;
    o[0] = z;
    o[1] = (zn - z) / (xk[xi + 1] - xk[xi]);
  }
}

template <class T>
struct EvalSlopes2Args {
  const T* xk;             // [nx]
  const T* yk;             // [ny]
  uint32_t nx, ny;
  DenseLut<T> dx, dy;
  const T* recs;           // point records [(nx-1)][ny][lanes][2] = {z, m}, or cell records (slope_pack_kernel)
  uint32_t row_bytes, col_bytes;    // byte offset of a query's first record = xi * row_bytes + yi * col_bytes
  int debug;               // NDI_TUNING builds only (measurement aid, results meaningless): bit 0 no record loads, bit 1 no
                           // stores, bit 2 no searches, bit 3 no division
  const T* qx;
  const T* qy;
  T* out;
  uint64_t nq, out_stride;
  uint32_t lanes;
  int mode;
  unsigned long long* first_fail;   // [2]: x, y
  int check;                        // see EvalLanesArgs
};

template <class T>
struct Pair2;
template <>
struct Pair2<double> { using type = dbl2; };
template <>
struct Pair2<float> { typedef float type __attribute__((ext_vector_type(2))); };

// the branch-free search with a compile-time number of knot reads per bucket (MK >= the index's maxk: knots behind the
// bucket's own -- and the +inf sentinels -- are > x and count nothing)
template <class T, int MK>
__device__ __forceinline__ uint32_t lane_axis_index_mk(const LaneAxis<T>& S, T x) {
  if (S.lut) {                                            // (workgroup-uniform)
    T f = (x - S.k0) * S.scale;
    f = fmax(f, T(0));                                    // below the axis, NaN -> bucket 0
    f = fmin(f, T(S.m - 1u));
    const uint32_t lo = S.lut[(uint32_t)f];
    const uint32_t cnt = lo + knots_le<T, MK>(S.k, lo, x);
    const uint32_t i = cnt ? cnt - 1u : 0u;
    return i < S.n - 2u ? i : S.n - 2u;
  }
  const T mm = S.gfac * (x - S.k0) + T(0);
  return (mm >= T(0)) ? (uint32_t)(mm < T(S.n - 2u) ? mm : T(S.n - 2u)) : 0u;
}

// Software pipeline of a wave (one batch = 64 consecutive queries, LC trips of 64 items).  The counter that orders a
// wave's memory operations (vmcnt) is in-order and counts stores: a load issued BEHIND a batch's stores can only be waited
// for by draining those stores -- microseconds per batch.  So every load is issued AHEAD of the stores it must not wait for:
//     iteration k:   evaluate batch k (its records were requested an iteration ago, its scalars are in strip k & 1)
//                    request the records of batch k + 1 (strip (k + 1) & 1 is complete) and the queries of batch k + 3
//                    store batch k                                  <- nothing later waits for these
//                    search batch k + 2 (queries requested an iteration ago) into strip k & 1
// The loop body is straight-line for full batches (unconditional stores: the compiler can count, `s_waitcnt vmcnt(N)` with
// N > 0); the one partial batch a wave may have is its last and is handled behind the loop.  The kernel's own range test
// (`check`) keeps the lowest failing index per lane in registers and publishes it once, at the end.
template <class T, int LC, int MK, int CONTIG, int TB>      // CONTIG: 0 strided rows, 1 contiguous, 2 contiguous through a
                                                           // wave-private LDS strip as 16-byte vectors (needs out 16-byte aligned)
__global__ __launch_bounds__(TB) void eval_slopes2d_kernel(EvalSlopes2Args<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using P2 = typename Pair2<T>::type;                // one {z, m} record / one pair of per-query scalars
  if (A.nq == 0) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr uint32_t WAVES = TB / 64;
  LaneAxis<T> SX, SY;
  XRecs<T> XY;
  size_t off = 0;
  SX = stage_lane_axis<T, TB>(smem_raw, off, A.xk, A.nx, A.dx);
  SY = stage_lane_axis<T, TB>(smem_raw, off, A.yk, A.ny, A.dy);
  XY = XRecs<T>{reinterpret_cast<typename XRecs<T>::U*>(smem_raw + off), A.ny - 1u};
  off += XRecs<T>::bytes(A.ny - 1u);
  stage_xrecs<T, TB>(XY, A.yk, A.ny);
  // per wave, twice: [64] {x - x1, y - y1} | [64] {dy, RN(1 / dy) or 0} | [64] record byte offsets
  constexpr uint32_t STRIP = 64u * (2u * (uint32_t)sizeof(P2) + 4u);
  unsigned char* const w_base = smem_raw + off + (size_t)wave * 2u * STRIP;
  auto strip_f = [&](uint32_t b) { return reinterpret_cast<P2*>(w_base + b * STRIP); };
  auto strip_d = [&](uint32_t b) { return reinterpret_cast<P2*>(w_base + b * STRIP + 64u * sizeof(P2)); };
  auto strip_o = [&](uint32_t b) { return reinterpret_cast<uint32_t*>(w_base + b * STRIP + 128u * sizeof(P2)); };
  constexpr int VN = Wide<T>::N;
  using V = typename VecT<T, VN>::type;
  T* const w_res = reinterpret_cast<T*>(smem_raw + off + (size_t)WAVES * 2u * STRIP) + (size_t)wave * 64u * LC;   // CONTIG == 2
  __syncthreads();
  const int check = A.check, mode = A.mode;
  unsigned long long limit = check ? NO_FAIL : (A.first_fail[0] < A.first_fail[1] ? A.first_fail[0] : A.first_fail[1]);
  if (limit > A.nq) limit = A.nq;
  // trip t serves items t * 64 + lane of the batch's 64 * LC: query and channel of this lane in every trip
  uint32_t tq[LC], tc[LC];
#pragma unroll
  for (int t = 0; t < LC; ++t) {
    const uint32_t it = (uint32_t)t * 64u + lane;
    tq[t] = it / (uint32_t)LC;
    tc[t] = (it - tq[t] * (uint32_t)LC) * (uint32_t)sizeof(P2);
  }
  const uint64_t wstep = (uint64_t)gridDim.x * TB;
  uint64_t base = ((uint64_t)blockIdx.x * WAVES + wave) * 64u;
  if (base >= limit) return;
  const char* __restrict__ const recs = reinterpret_cast<const char*>(A.recs);
  T* __restrict__ const out = A.out;
  const uint32_t row_bytes = A.row_bytes, col_bytes = A.col_bytes;
  unsigned long long failx = NO_FAIL, faily = NO_FAIL;   // (check) lowest failing query of this lane, per axis
  // one batch's searches: cell, (x - x1), (y - y1), the y spacing with its reciprocal -> strip b
  auto phase1 = [&](uint32_t b, uint64_t bbase, T x, T y) {
    if (check) {                                     // Interp2D::is_in_x_range / is_in_y_range, see lane_check2
      const bool badx = (mode == EX_NO) ? !((SX.k0 <= x) && (x <= SX.kn)) : !(x == x);
      const bool bady = (mode == EX_NO) ? !((SY.k0 <= y) && (y <= SY.kn)) : !(y == y);
      const unsigned long long qi = bbase + lane;
      if (badx && qi < limit && qi < failx) failx = qi;
      if (bady && qi < limit && qi < faily) faily = qi;
    }
    uint32_t xi = NDI_CHK((lane_axis_index_mk<T, MK>(SX, x)), SX.n - 1u, BC_CELL_X);
    uint32_t yi = NDI_CHK((lane_axis_index_mk<T, MK>(SY, y)), SY.n - 1u, BC_CELL_Y);
#ifdef NDI_TUNING
    if (A.debug & 4) { xi = ((uint32_t)(x * T(977)) * 7u + lane) % (SX.n - 1u); yi = ((uint32_t)(y * T(1013)) * 13u + lane) % (SY.n - 1u); }
#endif
    T ry[4];
    XY.get(yi, ry);
    P2 f, d;
    f[0] = x - SX.k[xi];                   // linear.rs:35's (x - x1) of both directions
    f[1] = y - ry[0];
    d[0] = ry[1];
    d[1] = ry[2];                          // 0: the divisor is outside the shared-divisor window
    strip_f(b)[lane] = f;
    strip_d(b)[lane] = d;
    strip_o(b)[lane] = xi * row_bytes + yi * col_bytes;
  };
  auto load_q = [&](uint64_t bbase, T& x, T& y) {
    const uint64_t pn = bbase + lane;
    const uint64_t pcq = pn < A.nq ? pn : A.nq - 1u;
#ifdef NDI_TUNING
    if (A.debug & 64) { x = SX.k0 + T(lane); y = SY.k0 + T(pcq & 63u); return; }
#endif
    x = A.qx[pcq];
    y = A.qy[pcq];
  };
  P2 r1[LC], r2[LC];
  auto request = [&](uint32_t b) {                   // the records of strip b's batch: all trips
    const uint32_t* const co = strip_o(b);
#pragma unroll
    for (int t = 0; t < LC; ++t) {
      const uint32_t o = co[NDI_CHK(tq[t], 64u, BC_STRIP)] + tc[t];
#ifdef NDI_TUNING
      if (A.debug & 1) { r1[t] = P2(T(o)); r2[t] = P2(T(o + 1u)); continue; }
#endif
      r1[t] = *reinterpret_cast<const P2*>(recs + o);
      r2[t] = *reinterpret_cast<const P2*>(recs + o + (uint32_t)LC * (uint32_t)sizeof(P2));
    }
  };
  T res[LC];
  auto evaluate = [&](uint32_t b) {                  // bilinear.rs:88-97 with the records' slopes
    const P2* const cf = strip_f(b);
    const P2* const cd = strip_d(b);
#pragma unroll
    for (int t = 0; t < LC; ++t) {
      P2 f = cf[tq[t]], d = cd[tq[t]];
#ifdef NDI_TUNING
      if (A.debug & 32) { f = P2(T(1.5)); d = P2(T(0.5)); }
#endif
      SharedDivisor<T> dy;
      dy.d = d[0]; dy.r = d[1]; dy.ok = d[1] > T(0);
      const T z1 = r1[t][1] * f[0] + r1[t][0];
      const T z2 = r2[t][1] * f[0] + r2[t][0];
      bool okg;
#ifdef NDI_TUNING
      if (A.debug & 8) { res[t] = (z2 - z1) * d[1] * f[1] + z1; continue; }
#endif
      T m = div_shared_fast<T>(z2 - z1, dy, okg);
      if (__builtin_expect(!okg, 0)) m = (z2 - z1) / d[0];       // outside the window: the IEEE division
      res[t] = m * f[1] + z1;
      if (CONTIG == 2) w_res[(uint32_t)t * 64u + lane] = res[t];
    }
  };
  T xa, ya, xb, yb;                                  // queries of the next two batches to be searched
  load_q(base, xa, ya);
  load_q(base + wstep, xb, yb);
  phase1(0u, base, xa, ya);
  __builtin_amdgcn_wave_barrier();
  request(0u);
  load_q(base + 2u * wstep, xa, ya);
  phase1(1u, base + wstep, xb, yb);
  __builtin_amdgcn_wave_barrier();
  uint32_t cur = 0;
  // one full batch.  (The first iteration is peeled so that the loop is entered in its steady state -- this batch's records
  // requested, the queries and the previous batch's stores behind them: the compiler takes the smaller count of two entry
  // paths, and a prologue without stores would make every iteration wait for part of the previous batch's stores.)
  auto full_batch = [&]() {
    evaluate(cur);
    load_q(base + 3u * wstep, xb, yb);
    request(cur ^ 1u);                               // batch k + 1: AHEAD of this batch's stores
#ifdef NDI_TUNING
    if ((A.debug & 2) && res[0] != T(-12345.678)) {} else
#endif
    if (CONTIG == 2) {                               // 64 * LC values = 64 * LC / VN vectors, lane after lane
      V* const ov = reinterpret_cast<V*>(out + base * (uint64_t)LC) + lane;
      const V* const sv = reinterpret_cast<const V*>(w_res) + lane;
      constexpr int NV = 64 * LC / VN;               // (64 * LC is a multiple of VN)
#pragma unroll
      for (int v = 0; v < (NV + 63) / 64; ++v)
        if (v * 64 + 64 <= NV || lane < (uint32_t)(NV - v * 64)) store_stream<true>(ov + v * 64, sv[v * 64]);
    } else if (CONTIG) {
      T* const orow = out + base * (uint64_t)LC + lane;
#ifdef NDI_TUNING
      if (A.debug & 16) {
#pragma unroll
        for (int t = 0; t < LC; ++t) orow[(uint32_t)t * 64u] = res[t];
      } else
#endif
#pragma unroll
      for (int t = 0; t < LC; ++t) store_stream<true>(orow + (uint32_t)t * 64u, res[t]);
    } else {
#pragma unroll
      for (int t = 0; t < LC; ++t) out[(base + tq[t]) * A.out_stride + tc[t] / (uint32_t)sizeof(P2)] = res[t];
    }
    __builtin_amdgcn_wave_barrier();                 // strip `cur` has been read: the batch after next takes it
    phase1(cur, base + 2u * wstep, xa, ya);
    xa = xb; ya = yb;
    __builtin_amdgcn_wave_barrier();
    base += wstep;
    cur ^= 1u;
  };
  if (limit - base >= 64u) {
    full_batch();
    while (base < limit && limit - base >= 64u) full_batch();
  }
  if (base < limit) {                                // the wave's last, partial batch
    const uint32_t nq_here = (uint32_t)(limit - base);
    evaluate(cur);
#pragma unroll
    for (int t = 0; t < LC; ++t)
      if (tq[t] < nq_here) {
        if (CONTIG) out[base * (uint64_t)LC + ((uint32_t)t * 64u + lane)] = res[t];
        else out[(base + tq[t]) * A.out_stride + tc[t] / (uint32_t)sizeof(P2)] = res[t];
      }
  }
  if (check) {
    if (failx != NO_FAIL) atomicMin(&A.first_fail[0], failx);
    if (faily != NO_FAIL) atomicMin(&A.first_fail[1], faily);
  }
}

// ndi_output_alloc: Array::zeros (interp1d/mod.rs:209) -- the buffer is filled with zeros by 16-byte non-temporal stores; the
// duration of this very fill is the allocator's measure of where the buffer landed in physical memory.
// The fill walks the buffer the way the long-row evaluation writes it -- whole rows of 32 KiB at scattered positions, a
// workgroup per row, 16-byte non-temporal stores -- because that is the pattern whose rate depends on the buffer: a plain
// sequential fill runs at the same 6.76 TB/s into every buffer, including those the evaluation kernel writes 25 % slower
// (profiles/r06_fill_vs_kernel.jsonl).  Row r of the walk is row (r * mult) mod nrows, mult coprime to nrows (host).
__global__ __launch_bounds__(BLOCK) void zero_fill_kernel(dbl2* p, uint64_t nvec, uint64_t nrows, uint64_t mult) {
  constexpr uint64_t ROWV = 2048;   // 16-byte vectors per 32 KiB row
  const dbl2 z = {0.0, 0.0};
  for (uint64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
    const uint64_t row = (r * mult) % nrows;
    dbl2* const o = p + row * ROWV;
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint64_t v = (uint64_t)u * BLOCK + threadIdx.x;
      if (row * ROWV + v < nvec) __builtin_nontemporal_store(z, o + v);
    }
  }
}

// Measurement aid (ndi_interp2d_probe_ceiling): the memory access mix of eval_bilinear_kernel and nothing else --
// per item one pre-generated uniformly random cell, the four corner vectors from the handle's own grid with the
// kernel's lane mapping, a token amount of arithmetic, the output vector stored -- no searches, no knots, no query
// values.  Its time is the ceiling the memory system sets for this gather on this box and this table.
template <class T, int VEC>
__global__ __launch_bounds__(BLOCK) void probe_gather_kernel(const T* data, uint64_t nx, uint64_t ny, uint64_t row_cells,
                                                             uint64_t cell_elems, uint64_t lanes, uint64_t nq,
                                                             uint64_t seed, T* out, uint64_t out_stride) {
  using V = typename VecT<T, VEC>::type;
  const uint32_t LV = (uint32_t)(lanes / VEC);
  const uint64_t items = nq * LV;
  const uint64_t ncx = nx - 1, ncy = ny - 1;
  for (uint64_t it = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; it < items; it += (uint64_t)gridDim.x * BLOCK) {
    const uint64_t q = it / LV;
    const uint32_t v = (uint32_t)(it - q * LV);
    uint64_t z = (q + seed) * 0x9E3779B97F4A7C15ull;   // splitmix64: the cell of query q
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const uint64_t xi = (z >> 32) % ncx, yi = (z & 0xffffffffu) % ncy;
    const V* z11 = reinterpret_cast<const V*>(data + (xi * row_cells + yi) * cell_elems);
    const V* z21 = reinterpret_cast<const V*>(data + ((xi + 1) * row_cells + yi) * cell_elems);
    const V a11 = z11[v], a12 = z11[LV + v], a21 = z21[v], a22 = z21[LV + v];
    const T tx = T(0.25), ty = T(0.75);
    const V z1 = (a21 - a11) * tx + a11;
    const V z2 = (a22 - a12) * tx + a12;
    reinterpret_cast<V*>(out + q * out_stride)[v] = (z2 - z1) * ty + z1;
  }
}

// 2-D counterpart of eval_small_kernel: both searches and the bilinear evaluation in one launch, one query per
// thread (lanes <= SMALL_LANES), output into a staging buffer owned by the host side.
template <class T>
struct EvalSmall2Args {
  Pyramid<T> px, py;
  const T* data;
  const T* qx;
  const T* qy;
  T* out;  // [nq][out_stride]
  uint64_t nq, out_stride, row_cells, cell_elems;
  uint32_t lanes;
  int mode;
  unsigned long long* first_fail;  // [2]
  int prechecked;                  // see EvalSmallArgs
  BucketIndex<T> bx, by;           // bucket indices staged behind the pyramids when non-null (large device batches)
  int sdiv;                        // 1: the three divisions of a value share the query's two correctly rounded
                                   //    reciprocals (div_shared: the same bits as the IEEE divisions)
};

template <class T>
__global__ __launch_bounds__(BLOCK) void eval_small2d_kernel(EvalSmall2Args<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const uint32_t tid = threadIdx.x;
  const uint32_t nxa = A.px.n + A.px.n1, nya = A.py.n + A.py.n1;
  lds_u16 lutx = nullptr, luty = nullptr;
  {
    T* sx = reinterpret_cast<T*>(smem_raw);
    T* sy = sx + nxa;
    for (uint32_t i = tid; i < nxa; i += BLOCK) sx[i] = A.px.lv0[i];
    for (uint32_t i = tid; i < nya; i += BLOCK) sy[i] = A.py.lv0[i];
    if (A.bx.lut || A.by.lut) {   // [x pyramid | y pyramid | x lut | y lut], as in locate2_kernel
      const size_t off = ((size_t)(nxa + nya) * sizeof(T) + 15u) & ~(size_t)15u;
      uint32_t* sl = reinterpret_cast<uint32_t*>(smem_raw + off);
      const uint32_t wx = A.bx.lut ? (A.bx.m + 2u) / 2u : 0u, wy = A.by.lut ? (A.by.m + 2u) / 2u : 0u;
      const uint32_t* srcx = reinterpret_cast<const uint32_t*>(A.bx.lut);
      const uint32_t* srcy = reinterpret_cast<const uint32_t*>(A.by.lut);
      for (uint32_t i = tid; i < wx; i += BLOCK) sl[i] = srcx[i];
      for (uint32_t i = tid; i < wy; i += BLOCK) sl[wx + i] = srcy[i];
      if (wx) lutx = (lds_u16)(smem_raw + off);
      if (wy) luty = (lds_u16)(smem_raw + off + (size_t)wx * 4u);
    }
  }
  __syncthreads();
  PyramidLds<T> PX, PY;
  PX.lv0 = (lds_ptr<T>)(smem_raw);
  PX.lv1 = PX.lv0 + A.px.n;
  PX.n = A.px.n; PX.n1 = A.px.n1; PX.levels = A.px.levels; PX.guess = A.px.guess; PX.block = A.px.block;
  PY.lv0 = PX.lv0 + nxa;
  PY.lv1 = PY.lv0 + A.py.n;
  PY.n = A.py.n; PY.n1 = A.py.n1; PY.levels = A.py.levels; PY.guess = A.py.guess; PY.block = A.py.block;
  const T x0 = PX.lv0[0], xn = PX.lv0[PX.n - 1], y0 = PY.lv0[0], yn = PY.lv0[PY.n - 1];
  const uint32_t lane = tid & 63u;
  const uint32_t L = A.lanes;
  unsigned long long limit = NO_FAIL;
  if (A.prechecked) limit = A.first_fail[0] < A.first_fail[1] ? A.first_fail[0] : A.first_fail[1];
  if (limit > A.nq) limit = A.nq;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK + (tid & ~63u); base < limit; base += (uint64_t)gridDim.x * BLOCK) {
    const uint64_t qi = base + lane;
    const bool active = qi < limit;
    const T x = active ? A.qx[qi] : x0;
    const T y = active ? A.qy[qi] : y0;
    const uint32_t xi = lutx ? locate_index_lut<T>(PX, lutx, A.bx.m, A.bx.scale, x0, xn, x)
                             : locate_index<T, lds_ptr<T>>(PX, x0, xn, x, lane);
    const uint32_t yi = luty ? locate_index_lut<T>(PY, luty, A.by.m, A.by.scale, y0, yn, y)
                             : locate_index<T, lds_ptr<T>>(PY, y0, yn, y, lane);
    if (!active) continue;
    const bool badx = (A.mode == EX_NO) ? !((x0 <= x) && (x <= xn)) : !(x == x);
    const bool bady = (A.mode == EX_NO) ? !((y0 <= y) && (y <= yn)) : !(y == y);
    if (badx && !A.prechecked) atomicMin(&A.first_fail[0], (unsigned long long)qi);
    if (bady && !A.prechecked) atomicMin(&A.first_fail[1], (unsigned long long)qi);
    if (badx || bady) continue;
    const T x1 = PX.lv0[xi], x2 = PX.lv0[xi + 1], y1 = PY.lv0[yi], y2 = PY.lv0[yi + 1];
    const T* z11 = A.data + ((uint64_t)xi * A.row_cells + yi) * A.cell_elems;
    const T* z12 = z11 + L;
    const T* z21 = A.data + ((uint64_t)(xi + 1) * A.row_cells + yi) * A.cell_elems;
    const T* z22 = z21 + L;
    T* o = A.out + qi * A.out_stride;
    if (A.sdiv) {   // one IEEE reciprocal per direction and query, three correctly rounded shared-divisor divisions per value
      const SharedDivisor<T> dx = shared_divisor<T>(x2 - x1), dy = shared_divisor<T>(y2 - y1);
      for (uint32_t l = 0; l < L; ++l) {
        const T z1 = frac_shared<T, T>(x1, z11[l], dx, z21[l], x);   // bilinear.rs:88-97
        const T z2 = frac_shared<T, T>(x1, z12[l], dx, z22[l], x);
        o[l] = frac_shared<T, T>(y1, z1, dy, z2, y);
      }
    } else {
      for (uint32_t l = 0; l < L; ++l) {
        const T z1 = frac_v<T, T>(x1, z11[l], x2, z21[l], x);
        const T z2 = frac_v<T, T>(x1, z12[l], x2, z22[l], x);
        o[l] = frac_v<T, T>(y1, z1, y2, z2, y);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// spline build: batched Thomas, one lane per thread
// ---------------------------------------------------------------------------------------------
template <class T>
struct BuildArgs {
  const T* data;  // [n][lanes]
  T* ca;          // [n-1][lanes]  (rows double as scratch for the eliminated rhs)
  T* cb;          // [n-1][lanes]  (periodic: scratch for k1)
  const T* x;     // [n] knots (FUSED build: dx / up are formed from them on the fly)
  T left_up0;        // up[0] (FUSED build: the plan's boundary-specific first entry)
  uint64_t up_len;   // entries of the upper diagonal (FUSED build)
  T up_last;         // its last entry (boundary-specific, like up[0])
  const T* dx;    // [n-1]
  const T* up;    // [m]
  const T* w;     // [m]
  const T* midp;  // [m]
  const T* k2;    // [m] periodic
  uint64_t n, lanes;
  int left_kind, right_kind;  // EndKind
  T left_val, right_val;
  T nkL_tmp1, nkL_d, nkR_tmp1, nkR_d, dx0_sq, dxl_sq;
  T per_den;
  StatusBlock* status;
  // BoundaryCondition::Individual (cubic_spline.rs:332-347, 370-403): per-lane end kinds / values.
  // The elimination factors depend on the LEFT end kind only (rows 1..n-2) and the last row on both,
  // so the host prepares 4 forward plans (NotAKnot, FirstDeriv, SecondDeriv, parabola-end) and 4x4 last rows.
  const uint8_t* lane_cls;   // left | right << 2   (EndKind, 3 = the n == 3 not-a-knot parabola rows)
  const T* lane_lval;
  const T* lane_rval;
  const T* w4;      // [4][n]
  const T* midp4;   // [4][n]
  const T* up0_4;   // [4]
  const T* wl;      // [4][4]  w[n-1]
  const T* midl;    // [4][4]  mid'[n-1]
  // blocked sweeps (narrow trailing axes, many knots): see spline_blocked_* below
  T* rfull;         // [n][lanes] scratch: rhs -> r' -> r'/mid' -> local k     (nullptr: the per-lane serial kernels)
  T* ends;          // [nblocks][lanes] value at each block's last (forward) / first (backward) row, zero carry in
  T* carry;         // [nblocks][lanes] carry into each block
  const T* fP;      // [n] prod_{j = block start .. i} (-w[j])
  const T* dco;     // [n] -up[i] / mid'[i]
  const T* bP;      // [n] prod_{j = i .. block end} dco[j]
  uint64_t S, nblocks;
  uint64_t rows;    // rows of the system being swept (n; n - 2 for the condensed periodic system)
  // the derivatives k themselves, [n][lanes] (nullptr: not kept).  Kept for table sets small enough that the
  // query-order kernel can hold {y, k} in LDS and re-form a / b per item (eval_fused_kernel, TLDS == 2).
  T* kout;
};

// SPLINE_GENERAL: rows 0 and n-1 from the boundary kinds (cubic_spline.rs:597-670), interior
// rows :456-471, thomas :678-721, then a/b :354-365 fused into the back substitution.
// PER_LANE: BoundaryCondition::Individual -- every lane selects its own end kinds / values and the
// matching precomputed elimination plan (arithmetic per lane identical to a scalar solve of that column,
// which is what solve_for_k_individual :370-403 does).
// SPLINE_GENERAL is two kernels.
//  spline_rhs_kernel    every right-hand side of the system (rows 0 .. n-1: boundary rows :597-670, interior rows
//                       :456-471) -- none depends on the recurrence, so this part is fully parallel, one thread per
//                       (row, lane).  Rows 0..n-2 go to the `a` table, row n-1 to the last row of the `b` table
//                       (both are scratch until the back substitution overwrites them).
//  spline_build_general_kernel
//                       the serial part, one lane per thread: thomas :678-721 (forward elimination of the
//                       right-hand sides with the shared factors, back substitution) fused with a/b :354-365.
//                       Rows are processed SB at a time with the next block's loads already in flight.
// Per element the operations and their order are exactly those of the reference's row-by-row form.
// PER_LANE: BoundaryCondition::Individual -- every lane selects its own end kinds / values and the matching
// precomputed elimination plan (what solve_for_k_individual :370-403 does one column at a time).
constexpr int SB = 16;

// dx[i] = x[i+1] - x[i] (i < n - 1) and the upper diagonal of the system being swept -- up[0] and up[len - 1] are
// boundary-specific scalars from the host plan, up[i] = x[i] - x[i-1] between them (cubic_spline.rs:440-451) -- formed on
// the device from the knots that are already there (the host plan states the same subtractions in T: host_logic.hpp).
template <class T>
__global__ __launch_bounds__(BLOCK) void spline_dx_up_kernel(const T* x, T* dx, T* up, uint64_t n, uint64_t up_len,
                                                             T up_first, T up_last) {
  for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
    if (i + 1 < n) dx[i] = x[i + 1] - x[i];
    if (i < up_len) up[i] = (i == 0) ? up_first : ((i + 1 == up_len) ? up_last : x[i] - x[i - 1]);
  }
}

// Right-hand side of row i, lane l (rows 0 and n-1: boundary rows :597-670, interior rows :456-471).  FROMX: the knot
// spacings are formed from the knots themselves (the same subtraction in T the dx array holds) -- the single-launch build
// of small systems has no dx array.
template <class T, bool PER_LANE, bool FROMX>
__device__ __forceinline__ T spline_rhs_at(const BuildArgs<T>& A, uint64_t i, uint64_t l) {
  const uint64_t n = A.n, L = A.lanes;
  const T two = T(2), three = T(3);
  auto dxa = [&](uint64_t j) -> T { return FROMX ? A.x[j + 1] - A.x[j] : A.dx[j]; };
  const T* y = A.data + l;
  int lk = A.left_kind, rk = A.right_kind;
  T lval = A.left_val, rval = A.right_val;
  if (PER_LANE) {
    const uint32_t cls = A.lane_cls[l];
    lk = (int)(cls & 3u);
    rk = (int)(cls >> 2);
    lval = A.lane_lval[l];
    rval = A.lane_rval[l];
  }
  T r;
  if (i == 0) {
    const T y0 = y[0], y1 = y[L], y2 = y[2 * L];
    const T dx0 = dxa(0), dx1 = dxa(1);
    if (lk == 0) r = (A.nkL_tmp1 * (y1 - y0) / dx0 + A.dx0_sq * (y2 - y1) / dx1) / A.nkL_d;
    else if (lk == 1) r = lval;
    else if (lk == 2) r = three * (y1 - y0) - lval * A.dx0_sq / two;
    else r = ((y1 - y0) / dx0) * two;                                  // parabola rows (:592), n == 3 only
  } else if (i + 1 == n) {
    const T ym = y[(n - 3) * L], yc = y[(n - 2) * L], yp = y[(n - 1) * L];
    const T dxl = dxa(n - 2), dxl2 = dxa(n - 3);
    if (rk == 0) r = (A.dxl_sq * (yc - ym) / dxl2 + A.nkR_tmp1 * (yp - yc) / dxl) / A.nkR_d;
    else if (rk == 1) r = rval;
    else if (rk == 2) r = three * (yp - yc) + rval * A.dxl_sq / two;
    else r = ((yp - yc) / dxl) * two;                                  // parabola rows (:595)
  } else {
    const T a0 = y[(i - 1) * L], a1 = y[i * L], a2 = y[(i + 1) * L];
    const T dxn = dxa(i), dxn_1 = dxa(i - 1);
    if (PER_LANE && lk == 3) r = (((a2 - a1) / dxa(1)) * dxa(0) + ((a1 - a0) / dxa(0)) * dxa(1)) * three;  // :593-594
    else r = three * (dxn * (a1 - a0) / dxn_1 + dxn_1 * (a2 - a1) / dxn);
  }
  return r;
}

template <class T, bool PER_LANE>
__global__ __launch_bounds__(BLOCK) void spline_rhs_kernel(BuildArgs<T> A) {
  const uint64_t n = A.n, L = A.lanes;
  const uint64_t total = n * L;
  for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (uint64_t)gridDim.x * BLOCK) {
    const uint64_t i = e / L, l = e - i * L;
    const T r = spline_rhs_at<T, PER_LANE, false>(A, i, l);
    if (A.rfull) A.rfull[e] = r;                 // blocked sweeps: one array for all n rows
    else if (i + 1 == n) A.cb[(n - 2) * L + l] = r;
    else A.ca[i * L + l] = r;
  }
}

// FUSED (the single-launch build of small systems: the reference's (100, 5), 1024 x 8, ...): no spline_rhs_kernel /
// spline_dx_up_kernel passes before this one -- the right-hand sides are formed here row by row (spline_rhs_at, the same
// operations), dx / up come from the knots; the eliminated right-hand sides still park in the `a` table.  Three launches,
// a temporary plan buffer and two stream-order dependencies become one launch.
template <class T, bool PER_LANE, bool KOUT = false, bool FUSED = false>
__global__ __launch_bounds__(64) void spline_build_general_kernel(BuildArgs<T> A) {
  const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= A.lanes) return;
  const uint64_t n = A.n, L = A.lanes;
  auto dx_at = [&](uint64_t i) -> T { return FUSED ? A.x[i + 1] - A.x[i] : A.dx[i]; };
  auto up_at = [&](uint64_t i) -> T {   // (i >= 1; row 0's entry is the plan's up0)
    if (!FUSED) return A.up[i];
    return (i + 1 == A.up_len) ? A.up_last : A.x[i] - A.x[i - 1];
  };
  auto rhs_row = [&](uint64_t i, const T* sa_row) -> T {
    if (FUSED) return spline_rhs_at<T, PER_LANE, true>(A, i, l);
    return *sa_row;
  };
  const T* y = A.data + l;
  T* sa = A.ca + l;
  T* sb = A.cb + l;
  const T* w = A.w;
  const T* midp = A.midp;
  T up0 = FUSED ? A.left_up0 : A.up[0], w_last = A.w[n - 1], mid_last = A.midp[n - 1];
  if (PER_LANE) {
    const uint32_t cls = A.lane_cls[l];
    const int lk = (int)(cls & 3u), rk = (int)(cls >> 2);
    w = A.w4 + (uint64_t)lk * n;
    midp = A.midp4 + (uint64_t)lk * n;
    up0 = A.up0_4[lk];
    w_last = A.wl[lk * 4 + rk];
    mid_last = A.midl[lk * 4 + rk];
  }
  const T rhs_last = FUSED ? spline_rhs_at<T, PER_LANE, true>(A, n - 1, l) : sb[(n - 2) * L];
  // The per-row factors (w, up, mid', dx) are fetched per block together with the rows, *before* the block's
  // stores: they cannot be scalar loads (the compiler cannot prove they do not alias the tables being written)
  // and a load placed after a store waits for that store (vmcnt is in-order and counts stores on CDNA4).
  // ---- forward elimination of the right-hand sides, rows 1 .. n-2 (row 0 stays as it is)
  T r_prev = rhs_row(0, sa);
  if (FUSED) sa[0] = r_prev;            // (the back substitution reads row 0's right-hand side from the table)
  T rn[SB], wn[SB];
  {
    const uint64_t left0 = n - 2;
    const int cnt0 = left0 < (uint64_t)SB ? (int)left0 : SB;
#pragma unroll
    for (int b = 0; b < SB; ++b) {
      if (b < cnt0) {
        rn[b] = rhs_row(1 + (uint64_t)b, sa + (1 + (uint64_t)b) * L);
        wn[b] = w[1 + b];
      }
    }
  }
  for (uint64_t i0 = 1; i0 + 1 < n; i0 += SB) {
    const uint64_t left = n - 1 - i0;
    const int cnt = left < (uint64_t)SB ? (int)left : SB;
    T rc[SB], wc[SB];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
      rc[b] = rn[b];
      wc[b] = wn[b];
    }
    {
      const uint64_t j0 = i0 + SB;   // next block, requested before this one is processed
      if (j0 + 1 < n) {
        const uint64_t leftn = n - 1 - j0;
        const int cntn = leftn < (uint64_t)SB ? (int)leftn : SB;
#pragma unroll
        for (int b = 0; b < SB; ++b) {
          if (b < cntn) {
            rn[b] = rhs_row(j0 + b, sa + (j0 + b) * L);
            wn[b] = w[j0 + b];
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < SB; ++b) {
      if (b < cnt) {
        const T r = rc[b] - wc[b] * r_prev;
        sa[(i0 + b) * L] = r;
        r_prev = r;
      }
    }
  }
  const T r_last = rhs_last - w_last * r_prev;
  T k_next = r_last / mid_last;
  if (KOUT) A.kout[(n - 1) * L + l] = k_next;
  T y_hi = y[(n - 1) * L];
  // ---- back substitution fused with a/b, rows n-2 .. 0, SB rows at a time, next block's loads in flight
  T ri_next[SB], yl_next[SB], up_next[SB], mid_next[SB], dx_next[SB];
  {
    const uint64_t hi0 = n - 1;
    const int cnt0 = hi0 < (uint64_t)SB ? (int)hi0 : SB;
#pragma unroll
    for (int b = 0; b < SB; ++b) {
      if (b < cnt0) {
        const uint64_t i = hi0 - 1 - b;
        ri_next[b] = sa[i * L];
        yl_next[b] = y[i * L];
        up_next[b] = (i == 0) ? up0 : up_at(i);
        mid_next[b] = midp[i];
        dx_next[b] = dx_at(i);
      }
    }
  }
  for (uint64_t hi = n - 1; hi > 0;) {   // rows hi-1, hi-2, ...
    const int cnt = hi < (uint64_t)SB ? (int)hi : SB;
    T ri[SB], yl[SB], upc[SB], midc[SB], dxc[SB];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
      ri[b] = ri_next[b];
      yl[b] = yl_next[b];
      upc[b] = up_next[b];
      midc[b] = mid_next[b];
      dxc[b] = dx_next[b];
    }
    {
      const uint64_t hn = hi - (uint64_t)cnt;   // the next block covers rows hn-1, hn-2, ...
      if (hn > 0) {
        const int cntn = hn < (uint64_t)SB ? (int)hn : SB;
#pragma unroll
        for (int b = 0; b < SB; ++b) {
          if (b < cntn) {
            const uint64_t i = hn - 1 - b;
            ri_next[b] = sa[i * L];
            yl_next[b] = y[i * L];
            up_next[b] = (i == 0) ? up0 : up_at(i);
            mid_next[b] = midp[i];
            dx_next[b] = dx_at(i);
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < SB; ++b) {
      if (b < cnt) {
        const uint64_t i = hi - 1 - b;
        const T k = (ri[b] - upc[b] * k_next) / midc[b];
        const T dy = y_hi - yl[b];
        sa[i * L] = k * dxc[b] - dy;
        sb[i * L] = dy - k_next * dxc[b];
        if (KOUT) A.kout[i * L + l] = k;
        k_next = k;
        y_hi = yl[b];
      }
    }
    hi -= (uint64_t)cnt;
  }
}

// WIDE trailing axes (round 6; BASELINE configs[1]: 4096 knots x 4096 lanes): the serial kernel above runs ONE wave per 64
// lanes -- 64 waves on 256 CUs at C2 -- and that wave does everything: it waits for its own loads (16 rows in flight),
// and a separate pass has to write every right-hand side to memory first and read it back (1.06 GB moved for 0.40 GB of
// data in / tables out at 0.05 of the HBM peak: profiles/r05_pmc_hbm_counters.txt).  Here a workgroup of four waves owns
// the 64 lanes: waves 1-3 PRODUCE a block of RB rows ahead -- forward: the right-hand sides (spline_rhs_at, the same
// function, two divisions per element, no serial dependency) straight into LDS, never to memory; backward: the eliminated
// right-hand sides and the data rows, and per row up / mid' / RN(1 / mid') / dx -- while wave 0 CONSUMES the previous
// block out of LDS: the recurrence alone (cubic_spline.rs:690-702, 711-720) and the a / b epilogue (:354-365), one
// barrier per block; every wave requests a whole block's operands before it computes (RW rows per producer wave).  Per element the operations and their order are those of spline_build_general_kernel; the back
// substitution's division by mid'[i] -- one divisor for all lanes of a row -- is the correctly rounded shared-divisor
// division (div_shared_fast: the bits of the IEEE division, which is redone when an operand leaves the exponent window).
// Bit-identical tables (test_spline_coefficients_bit_exact runs both kernels).  LDS: 2 x RB x 64 x 2 values + factors.
template <class T, int RW>      // RW rows per producer wave and block; a block is RB = 3 RW rows
__global__ __launch_bounds__(256) void spline_build_wide_kernel(BuildArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int RB = 3 * RW;
  T* const s_r = reinterpret_cast<T*>(smem_raw);                       // [2][RB][64]
  T* const s_y = s_r + 2 * RB * 64;                                    // [2][RB][64]
  T* const s_f = s_y + 2 * RB * 64;                                    // [2][4][RB]
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint64_t n = A.n, L = A.lanes;
  const uint64_t l0 = (uint64_t)blockIdx.x * 64u, l = l0 + lane;
  const bool live = l < L;
  const uint64_t lc = live ? l : L - 1u;                               // (loads of the padding lanes stay inside the rows)
  const T* __restrict__ const y = A.data;
  T* const sa = A.ca;
  T* const sb = A.cb;
  const T three = T(3);
  // ---- forward elimination, rows 1 .. n-2 (row 0's right-hand side stays as it is)
  const uint64_t nblk = (n - 2 + RB - 1) / RB;
  // producer wave pw (0 .. 2): interior rows i0 + pw RW .. + RW - 1 of this lane: RW + 2 data rows and RW + 1 spacings
  // requested together, then the right-hand sides (cubic_spline.rs:456-471, spline_rhs_at's interior branch verbatim)
  auto produce_fwd = [&](uint32_t buf, uint64_t i0) {
    const uint32_t pw = wave - 1u;
    const uint64_t r0 = i0 + (uint64_t)pw * RW;                        // first row of this wave
    T yv[RW + 2], dv[RW + 1];
#pragma unroll
    for (int k = 0; k < RW + 2; ++k) {
      const uint64_t row = r0 - 1 + (uint64_t)k;
      yv[k] = y[(row < n ? row : n - 1) * L + lc];
    }
#pragma unroll
    for (int k = 0; k < RW + 1; ++k) {
      const uint64_t row = r0 - 1 + (uint64_t)k;
      dv[k] = A.dx[row < n - 1 ? row : n - 2];
    }
    T* const r = s_r + (size_t)buf * RB * 64 + (size_t)pw * RW * 64;
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const T a0 = yv[k], a1 = yv[k + 1], a2 = yv[k + 2];
      const T dxn = dv[k + 1], dxn_1 = dv[k];
      r[k * 64 + lane] = three * (dxn * (a1 - a0) / dxn_1 + dxn_1 * (a2 - a1) / dxn);   // (rows beyond n-2: never consumed)
    }
    if (lane < (uint32_t)RW) {
      const uint64_t row = r0 + lane;
      s_f[(size_t)buf * 4 * RB + pw * RW + lane] = A.w[row < n ? row : n - 1];
    }
  };
  T r_prev = T(0);
  if (wave == 0) {
    r_prev = spline_rhs_at<T, false, false>(A, 0, lc);
    if (live) sa[l] = r_prev;
  } else if (nblk) {
    produce_fwd(0u, 1u);
  }
  __syncthreads();
  for (uint64_t blk = 0; blk < nblk; ++blk) {
    const uint64_t i0 = 1u + blk * RB;
    const uint32_t buf = (uint32_t)(blk & 1u);
    if (wave != 0) {
      if (blk + 1 < nblk) produce_fwd(buf ^ 1u, i0 + RB);
    } else {
      const uint64_t left = n - 1 - i0;
      const T* const r = s_r + (size_t)buf * RB * 64;
      const T* const wf = s_f + (size_t)buf * 4 * RB;
      if (left >= (uint64_t)RB) {                                      // a full block: operands first, then the chain
        T rv[RB], wv[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          rv[b] = r[b * 64 + lane];
          wv[b] = wf[b];
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          const T rr = rv[b] - wv[b] * r_prev;                         // cubic_spline.rs:699-701
          rv[b] = rr;
          r_prev = rr;
        }
        if (live) {
#pragma unroll
          for (int b = 0; b < RB; ++b) sa[(i0 + (uint64_t)b) * L + l] = rv[b];
        }
      } else {
        for (int b = 0; b < (int)left; ++b) {
          const T rr = r[b * 64 + lane] - wf[b] * r_prev;
          if (live) sa[(i0 + (uint64_t)b) * L + l] = rr;
          r_prev = rr;
        }
      }
    }
    __syncthreads();
  }
  // ---- last row, then the back substitution fused with a / b, rows n-2 .. 0
  const uint64_t bblk = (n - 1 + RB - 1) / RB;
  auto produce_bwd = [&](uint32_t buf, uint64_t hi) {                  // rows hi-1, hi-2, ..., at most RB of them
    const uint32_t pw = wave - 1u;
    T* const r = s_r + (size_t)buf * RB * 64 + (size_t)pw * RW * 64;
    T* const yy = s_y + (size_t)buf * RB * 64 + (size_t)pw * RW * 64;
    T rv[RW], yv[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const uint64_t b = (uint64_t)pw * RW + (uint64_t)k;
      const uint64_t i = b < hi ? hi - 1 - b : 0;
      rv[k] = sa[i * L + lc];
      yv[k] = y[i * L + lc];
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      r[k * 64 + lane] = rv[k];
      yy[k * 64 + lane] = yv[k];
    }
    if (lane < (uint32_t)RW) {
      const uint64_t b = (uint64_t)pw * RW + lane;
      const uint64_t i = b < hi ? hi - 1 - b : 0;
      T* const f = s_f + (size_t)buf * 4 * RB;
      const SharedDivisor<T> sd = shared_divisor<T>(A.midp[i]);
      f[0 * RB + b] = A.up[i];
      f[1 * RB + b] = sd.d;
      f[2 * RB + b] = sd.ok ? sd.r : T(0);
      f[3 * RB + b] = A.dx[i];
    }
  };
  T k_next = T(0), y_hi = T(0);
  if (wave == 0) {
    const T rhs_last = spline_rhs_at<T, false, false>(A, n - 1, lc);
    const T r_last = rhs_last - A.w[n - 1] * r_prev;
    k_next = r_last / A.midp[n - 1];                                   // :704-708
    y_hi = y[(n - 1) * L + lc];
  } else {
    produce_bwd(0u, n - 1);
  }
  __syncthreads();
  auto back_row = [&](uint64_t i, T ri, T yl, T up, T mid, T rmid, T dxi) {
    SharedDivisor<T> sd;
    sd.d = mid; sd.r = rmid; sd.ok = rmid > T(0);
    const T num = ri - up * k_next;
    bool ok;
    T k = div_shared_fast<T>(num, sd, ok);
    if (__builtin_expect(!ok, 0)) k = num / mid;                       // :716-718
    const T dy = y_hi - yl;
    if (live) {
      sa[i * L + l] = k * dxi - dy;                                    // :354-365
      sb[i * L + l] = dy - k_next * dxi;
    }
    k_next = k;
    y_hi = yl;
  };
  for (uint64_t blk = 0; blk < bblk; ++blk) {
    const uint64_t hi = n - 1 - blk * RB;
    const uint32_t buf = (uint32_t)(blk & 1u);
    if (wave != 0) {
      if (blk + 1 < bblk) produce_bwd(buf ^ 1u, hi - RB);
    } else {
      const T* const r = s_r + (size_t)buf * RB * 64;
      const T* const yy = s_y + (size_t)buf * RB * 64;
      const T* const f = s_f + (size_t)buf * 4 * RB;
      if (hi >= (uint64_t)RB) {
#pragma unroll
        for (int h = 0; h < 3; ++h) {                                  // RW rows at a time: operands first, then the chain
          T rv[RW], yv[RW], fu[RW], fm[RW], fr[RW], fd[RW];
#pragma unroll
          for (int k = 0; k < RW; ++k) {
            const int b = h * RW + k;
            rv[k] = r[b * 64 + lane]; yv[k] = yy[b * 64 + lane];
            fu[k] = f[0 * RB + b]; fm[k] = f[1 * RB + b]; fr[k] = f[2 * RB + b]; fd[k] = f[3 * RB + b];
          }
#pragma unroll
          for (int k = 0; k < RW; ++k) back_row(hi - 1 - (uint64_t)(h * RW + k), rv[k], yv[k], fu[k], fm[k], fr[k], fd[k]);
        }
      } else {
        for (int b = 0; b < (int)hi; ++b)
          back_row(hi - 1 - (uint64_t)b, r[b * 64 + lane], yy[b * 64 + lane], f[0 * RB + b], f[1 * RB + b], f[2 * RB + b], f[3 * RB + b]);
      }
    }
    __syncthreads();
  }
}

// The whole build of a SMALL system in one workgroup (the reference's own bench shape (100, 5), benches/bench_interp1d.rs:
// 82-86: 500 values): data and right-hand sides live in LDS.  Phase A, all threads: every right-hand side (spline_rhs_at: the
// boundary rows :597-670 and the interior rows :456-471, two divisions each) -- the part of the work that has no serial
// dependency, spread over the whole workgroup instead of the handful of lanes the trailing axis has.  Phase B, one thread
// per lane of the trailing axis: thomas (:678-721) and a / b (:354-365) out of LDS.  The per-lane serial kernel took 70 us
// for (100, 5) -- one wave, five live lanes, every row's two divisions and its loads in the dependent chain; this one ~10.
// Operations and their order per element are those of the serial kernels: bit-identical tables.
// LDS: [y (n * lanes) | rhs (n * lanes)]; lanes <= blockDim.x.
template <class T, bool KOUT>
__global__ __launch_bounds__(BLOCK) void spline_build_lds_kernel(BuildArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const uint32_t n = (uint32_t)A.n, L = (uint32_t)A.lanes, total = n * L;
  T* s_y = reinterpret_cast<T*>(smem_raw);
  T* s_r = s_y + total;
  for (uint32_t e = threadIdx.x; e < total; e += BLOCK) {
    const uint32_t i = e / L, l = e - i * L;
    s_y[e] = A.data[e];
    s_r[e] = spline_rhs_at<T, false, true>(A, i, l);
  }
  __syncthreads();
  const uint32_t l = threadIdx.x;
  if (l >= L) return;
  auto dx_at = [&](uint32_t i) -> T { return A.x[i + 1] - A.x[i]; };
  // forward elimination of the right-hand sides, rows 1 .. n-2 (row 0 stays as it is); UB rows per trip: their
  // operands are fetched together, only the one multiply-subtract per row is in the dependent chain
  constexpr int UB = 8;
  T r_prev = s_r[l];
  for (uint32_t i0 = 1; i0 + 1 < n; i0 += UB) {
    T rv[UB], wv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const uint32_t i = i0 + u < n - 1u ? i0 + u : n - 2u;        // (clamped: loads unconditional)
      rv[u] = s_r[i * L + l];
      wv[u] = const_load(A.w, i);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (i0 + u + 1 < n) {
        const T r = rv[u] - wv[u] * r_prev;
        s_r[(i0 + u) * L + l] = r;
        r_prev = r;
      }
    }
  }
  const T r_last = s_r[(n - 1) * L + l] - const_load(A.w, n - 1) * r_prev;
  T k_next = r_last / const_load(A.midp, n - 1);
  if (KOUT) A.kout[(uint64_t)(n - 1) * L + l] = k_next;
  T y_hi = s_y[(n - 1) * L + l];
  for (uint32_t hi = n - 1; hi > 0;) {      // back substitution fused with a / b, rows hi-1, hi-2, ..., UB per trip
    const uint32_t cnt = hi < (uint32_t)UB ? hi : (uint32_t)UB;
    T rv[UB], upv[UB], midv[UB], ylv[UB], dxv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const uint32_t i = (uint32_t)u < cnt ? hi - 1u - u : 0u;     // (clamped)
      rv[u] = s_r[i * L + l];
      ylv[u] = s_y[i * L + l];
      upv[u] = (i == 0) ? A.left_up0 : ((uint64_t)i + 1 == A.up_len ? A.up_last : A.x[i] - A.x[i - 1]);
      midv[u] = const_load(A.midp, i);
      dxv[u] = dx_at(i);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if ((uint32_t)u < cnt) {
        const uint32_t i = hi - 1u - u;
        const T k = (rv[u] - upv[u] * k_next) / midv[u];
        const T dy = y_hi - ylv[u];
        A.ca[(uint64_t)i * L + l] = k * dxv[u] - dy;
        A.cb[(uint64_t)i * L + l] = dy - k_next * dxv[u];
        if (KOUT) A.kout[(uint64_t)i * L + l] = k;
        k_next = k;
        y_hi = ylv[u];
      }
    }
    hi -= cnt;
  }
}

// The x-only elimination factors of the GENERAL system on the device, for axes of 1e5-1e6 knots where the host's division
// chain of n steps was half of the build (8.7 of 18 ms at 1e6 knots): w[i] = low[i] / mid'[i-1], mid'[i] = mid[i] - w[i] up[i-1]
// (thomas, cubic_spline.rs:690-692, on the diagonals alone) with low / mid / up formed from the knots (:440-451).  A chain
// started a few rows early from the uneliminated diagonal forgets its start: an error e in mid'[i-1] becomes
// e low[i] up[i-1] / mid'[i-1]^2 in mid'[i] -- not bounded in absolute terms where dx[i] is much larger than its neighbours,
// but the RELATIVE error contracts: (e'/mid'[i]) / (e/mid'[i-1]) = low[i] up[i-1] / (mid'[i-1] mid'[i]) =
// dx[i] dx[i-2] / (mid'[i-1] mid'[i]) <= 9/16, because mid'[i] >= 4/3 (dx[i] + dx[i-1]) on every strictly rising axis (by
// induction: mid'[i] = 2 (dx[i] + dx[i-1]) - dx[i] dx[i-2] / mid'[i-1] >= 2 dx[i] + 2 dx[i-1] - 3/4 dx[i]); evenly spaced knots
// contract by 0.07 per row.  WARM = 64 rows bring the worst case to (9/16)^64 ~ 1e-16 of the starting error (a relative
// 0.25 at most): every thread owns SE consecutive rows and runs the reference's recurrence, in the reference's operation
// order, from 64 rows before them (the first thread from row 0, exactly).  Used by the blocked build only -- whose tables
// carry their own few-ulp tolerance -- and only on axes that pass its `tame` test (forced blocked builds of other axes take
// the host's factors).
template <class T>
__global__ __launch_bounds__(BLOCK) void spline_eliminate_kernel(const T* x, uint64_t n, T up_first, T mid_first, T low_last,
                                                                 T mid_last, T* w, T* midp, uint64_t SE) {
  constexpr uint64_t WARM = 64;
  const T two = T(2);
  const uint64_t b = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
  const uint64_t i0 = b * SE;
  if (i0 >= n) return;
  const uint64_t i1 = (i0 + SE < n) ? i0 + SE : n;
  auto up_of = [&](uint64_t i) -> T { return i == 0 ? up_first : x[i] - x[i - 1]; };          // (rows < n - 1)
  auto mid_of = [&](uint64_t i) -> T {
    if (i == 0) return mid_first;
    if (i + 1 == n) return mid_last;
    return two * ((x[i + 1] - x[i]) + (x[i] - x[i - 1]));
  };
  auto low_of = [&](uint64_t i) -> T { return i + 1 == n ? low_last : x[i + 1] - x[i]; };     // (rows >= 1)
  uint64_t j = i0 > WARM ? i0 - WARM : 0;
  T mp = mid_of(j);                       // row j: exact when j == 0 (mid'[0] = mid[0]), a starting guess otherwise
  if (j >= i0) {
    w[j] = T(0);
    midp[j] = mp;
  }
  for (uint64_t i = j + 1; i < i1; ++i) {
    const T wi = low_of(i) / mp;
    mp = mid_of(i) - wi * up_of(i - 1);
    if (i >= i0) {
      w[i] = wi;
      midp[i] = mp;
    }
  }
}

// BLOCKED Thomas sweeps for narrow trailing axes with many knots (scalar data on 1e5-1e6 knots; 8 lanes on 4096).
// The per-lane serial kernel above gives such a build ONE wave with a handful of live lanes doing 2n dependent steps
// (a division in every step of the back substitution): 10x-100x behind one CPU core.  Both sweeps are first-order
// linear recurrences with lane-independent coefficients (spline_blocked_coef_kernel), so the rows are cut into
// blocks of S: every (block, lane) pair solves its block from a zero carry (spline_blocked_local_kernel, fully
// parallel), one thread per lane chains the nblocks carries (spline_blocked_carry_kernel), and every row is corrected
// by carry x (product of the coefficients between the block edge and the row).  The back substitution is evaluated
// as k[i] = r'[i] / mid'[i] + (-up[i] / mid'[i]) k[i+1] -- the reference has (r'[i] - up[i] k[i+1]) / mid'[i]
// (cubic_spline.rs:711-720).  This is THE ONE PATH WHOSE RESULTS ARE NOT BIT-IDENTICAL to the reference order: the
// coefficient tables agree with the oracle to a few ulp of the largest table entry (bar: 1e-10 f64 / 1e-5 f32,
// tests/test_gpu_spline_blocked.py); NDI_SPLINE_BLOCKED=0 keeps the serial kernels.
// The lane-independent coefficient products of the blocked sweeps, one thread per block of S rows:
//     fP[i] = prod_{j = block start .. i} (-w[j])      dco[i] = -up[i] / mid'[i]      bP[i] = prod_{j = i .. block end} dco[j]
// (w, up, mid' are the x-only factors of the host plan, host_logic.hpp).
template <class T>
__device__ __forceinline__ void spline_blocked_coef_block(const T* w, const T* up, const T* midp, T* fP, T* dco, T* bP,
                                                          uint64_t n, uint64_t S, uint64_t b);

template <class T>
__global__ __launch_bounds__(BLOCK) void spline_blocked_coef_kernel(const T* w, const T* up, const T* midp, T* fP,
                                                                    T* dco, T* bP, uint64_t n, uint64_t S,
                                                                    uint64_t nblocks) {
  const uint64_t b = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (b >= nblocks) return;
  spline_blocked_coef_block<T>(w, up, midp, fP, dco, bP, n, S, b);
}

template <class T>
__device__ __forceinline__ void spline_blocked_coef_block(const T* w, const T* up, const T* midp, T* fP, T* dco, T* bP,
                                                          uint64_t n, uint64_t S, uint64_t b) {
  const uint64_t i0 = b * S, i1 = (i0 + S < n) ? i0 + S : n;
  T p = T(0);
  for (uint64_t i = i0; i < i1; ++i) {
    const T c = -w[i];
    p = (i == i0) ? c : p * c;
    fP[i] = p;
  }
  for (uint64_t i = i1; i-- > i0;) {
    const T d = (i + 1 < n) ? -(up[i] / midp[i]) : T(0);
    dco[i] = d;
    p = (i + 1 == i1) ? d : d * p;
    bP[i] = p;
  }
}

template <class T, bool CL = true>
__device__ __forceinline__ void spline_blocked_local_task(const BuildArgs<T>& A, int backward, uint64_t t);

template <class T>
__global__ __launch_bounds__(BLOCK) void spline_blocked_local_kernel(BuildArgs<T> A, int backward) {
  const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;   // = block * L + lane
  if (t >= A.nblocks * A.lanes) return;
  spline_blocked_local_task<T>(A, backward, t);
}

template <class T, bool CL>
__device__ __forceinline__ void spline_blocked_local_task(const BuildArgs<T>& A, int backward, uint64_t t) {
  const uint64_t L = A.lanes, n = A.rows;
  const uint64_t b = t / L, l = t - b * L;
  const uint64_t i0 = b * A.S;
  const uint64_t i1 = (i0 + A.S < n) ? i0 + A.S : n;
  T* r = A.rfull + l;
  T prev = T(0);
  constexpr int UB = 8;   // rows per trip: their loads are independent of the recurrence and issued together
  if (!backward) {        // r'[i] = rhs[i] - w[i] r'[i-1]      (thomas, cubic_spline.rs:690-702)
    for (uint64_t i = i0; i < i1; i += UB) {
      T rv[UB], wv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (i + u < i1) { rv[u] = r[(i + u) * L]; wv[u] = const_load(A.w, i + u); }   // (w: the host plan's, in every form)
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (i + u < i1) { prev = rv[u] - wv[u] * prev; r[(i + u) * L] = prev; }
    }
  } else {                // k[i] = c[i] + d[i] k[i+1],  c = r' / mid' (already divided), from the block's last row down
    for (uint64_t i = i1; i > i0;) {
      const uint64_t cnt = (i - i0 < (uint64_t)UB) ? i - i0 : (uint64_t)UB;
      T cv[UB], dv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if ((uint64_t)u < cnt) { cv[u] = r[(i - 1 - u) * L]; dv[u] = ro_load<CL>(A.dco, i - 1 - u); }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if ((uint64_t)u < cnt) { prev = cv[u] + dv[u] * prev; r[(i - 1 - u) * L] = prev; }
      i -= cnt;
    }
  }
  A.ends[b * L + l] = prev;
}

template <class T, bool CL = true>
__device__ __forceinline__ void spline_blocked_carry_lane(const BuildArgs<T>& A, int backward, uint64_t l);

template <class T>
__global__ __launch_bounds__(64) void spline_blocked_carry_kernel(BuildArgs<T> A, int backward) {
  const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= A.lanes) return;
  spline_blocked_carry_lane<T>(A, backward, l);
}

template <class T, bool CL>
__device__ __forceinline__ void spline_blocked_carry_lane(const BuildArgs<T>& A, int backward, uint64_t l) {
  const uint64_t L = A.lanes, n = A.rows, nb = A.nblocks;
  T c = T(0);
  if (!backward) {
    for (uint64_t b = 0; b < nb; ++b) {
      A.carry[b * L + l] = c;
      const uint64_t last = ((b + 1) * A.S < n ? (b + 1) * A.S : n) - 1;
      c = A.ends[b * L + l] + ro_load<CL>(A.fP, last) * c;
    }
  } else {
    for (uint64_t b = nb; b-- > 0;) {
      A.carry[b * L + l] = c;
      c = A.ends[b * L + l] + ro_load<CL>(A.bP, b * A.S) * c;
    }
  }
}

// forward correction fused with the division of the back substitution: rfull[i] = (r'_local[i] + fP[i] carry) / mid'[i]
template <class T, bool CL = true>
__device__ __forceinline__ void spline_blocked_fix_forward_elem(const BuildArgs<T>& A, uint64_t e) {
  const uint64_t L = A.lanes;
  const uint64_t i = e / L, l = e - i * L;
  const T rp = A.rfull[e] + ro_load<CL>(A.fP, i) * A.carry[(i / A.S) * L + l];
  A.rfull[e] = rp / const_load(A.midp, i);
}
template <class T>
__global__ __launch_bounds__(BLOCK) void spline_blocked_fix_forward_kernel(BuildArgs<T> A) {
  const uint64_t total = A.rows * A.lanes;
  for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (uint64_t)gridDim.x * BLOCK)
    spline_blocked_fix_forward_elem<T>(A, e);
}

// backward correction fused with a / b (cubic_spline.rs:354-365):  k[i] = k_local[i] + bP[i] carry
template <class T, bool CL = true>
__device__ __forceinline__ void spline_blocked_finish_elem(const BuildArgs<T>& A, uint64_t e) {
  const uint64_t L = A.lanes;
  const uint64_t i = e / L, l = e - i * L;
  const T k0 = A.rfull[e] + ro_load<CL>(A.bP, i) * A.carry[(i / A.S) * L + l];
  const T k1 = A.rfull[e + L] + ro_load<CL>(A.bP, i + 1) * A.carry[((i + 1) / A.S) * L + l];
  const T dy = A.data[e + L] - A.data[e];
  const T dxi = ro_load<CL>(A.dx, i);
  A.ca[e] = k0 * dxi - dy;
  A.cb[e] = dy - k1 * dxi;
  if (A.kout) {
    A.kout[e] = k0;
    if (i + 2 == A.n) A.kout[e + L] = k1;
  }
}
template <class T>
__global__ __launch_bounds__(BLOCK) void spline_blocked_finish_kernel(BuildArgs<T> A) {
  const uint64_t total = (A.n - 1) * A.lanes;
  for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (uint64_t)gridDim.x * BLOCK)
    spline_blocked_finish_elem<T>(A, e);
}


// ---- periodic boundary (cubic_spline.rs:498-565) with the blocked sweeps: the condensed system of order m = n - 2
// is swept exactly as above (rows = m), then k = k1 + k_{n-2} k2 with the lane-independent k2 of the host plan.
// Right-hand sides of the condensed system, one thread per (row, lane); also the y[0] == y[n-1] check (:501-507).
template <class T>
__global__ __launch_bounds__(BLOCK) void spline_periodic_rhs_kernel(BuildArgs<T> A) {
  const uint64_t n = A.n, L = A.lanes, m = n - 2;
  const T three = T(3);
  const uint64_t total = m * L;
  for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (uint64_t)gridDim.x * BLOCK) {
    const uint64_t i = e / L, l = e - i * L;
    const T* y = A.data + l;
    T r;
    if (i == 0) {
      const T y0 = y[0], y1 = y[L], yn1 = y[(n - 1) * L], yn2 = y[(n - 2) * L];
      if (y0 != yn1) atomicAdd(&A.status->periodic_mismatch, 1ull);
      const T dx0 = A.dx[0], dxl = A.dx[n - 2];
      const T slope0 = (y1 - y0) / dx0;
      const T slope_1 = (yn1 - yn2) / dxl;
      r = (slope_1 * dx0 + slope0 * dxl) * three;
    } else {
      const T a0 = y[(i - 1) * L], a1 = y[i * L], a2 = y[(i + 1) * L];
      const T dxn = A.dx[i], dxn_1 = A.dx[i - 1];
      r = three * (dxn * (a1 - a0) / dxn_1 + dxn_1 * (a2 - a1) / dxn);
    }
    A.rfull[e] = r;
  }
}

// k_{n-2} per lane (:537-547) from the corrected k1[0] and k1[m-1]; stored in ends[lane]
template <class T>
__global__ __launch_bounds__(64) void spline_periodic_km1_kernel(BuildArgs<T> A) {
  const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= A.lanes) return;
  const uint64_t n = A.n, L = A.lanes, m = n - 2;
  const T three = T(3);
  const T* y = A.data + l;
  const T dxl = A.dx[n - 2], dxl2 = A.dx[n - 3];
  const T yn1 = y[(n - 1) * L], yn2 = y[(n - 2) * L], yn3 = y[(n - 3) * L];
  const T slope_1 = (yn1 - yn2) / dxl;
  const T slope_2 = (yn2 - yn3) / dxl2;
  const T rhs_last = (slope_2 * dxl + slope_1 * dxl2) * three;
  const T k1_first = A.rfull[l] + A.bP[0] * A.carry[l];
  const T k1_last = A.rfull[(m - 1) * L + l] + A.bP[m - 1] * A.carry[((m - 1) / A.S) * L + l];
  A.ends[l] = (rhs_last - k1_first * dxl2 - k1_last * dxl) / A.per_den;
}

// a / b (:354-365) with k[i] = k1[i] + k_{n-2} k2[i] (i < m), k[m] = k_{n-2}, k[n-1] = k[0]
template <class T>
__global__ __launch_bounds__(BLOCK) void spline_periodic_finish_kernel(BuildArgs<T> A) {
  const uint64_t n = A.n, L = A.lanes, m = n - 2, total = (n - 1) * L;
  for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (uint64_t)gridDim.x * BLOCK) {
    const uint64_t i = e / L, l = e - i * L;
    const T km1 = A.ends[l];
    auto kval = [&](uint64_t j) -> T {
      if (j == n - 1) j = 0;
      if (j == m) return km1;
      const T k1 = A.rfull[j * L + l] + const_load(A.bP, j) * A.carry[(j / A.S) * L + l];
      return k1 + km1 * const_load(A.k2, j);
    };
    const T k0 = kval(i), k1v = kval(i + 1);
    const T dy = A.data[e + L] - A.data[e];
    const T dxi = const_load(A.dx, i);
    A.ca[e] = k0 * dxi - dy;
    A.cb[e] = dy - k1v * dxi;
    if (A.kout) {
      A.kout[e] = k0;
      if (i + 2 == n) A.kout[e + L] = k1v;
    }
  }
}

// n == 3 closed forms: parabola (:569-596) and periodic (:480-496).
template <class T>
__global__ __launch_bounds__(64) void spline_build_n3_kernel(BuildArgs<T> A, int periodic) {
  const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= A.lanes) return;
  const uint64_t L = A.lanes;
  const T one = T(1), two = T(2), three = T(3);
  const T y0 = A.data[l], y1 = A.data[L + l], y2 = A.data[2 * L + l];
  const T dx0 = A.dx[0], dx1 = A.dx[1];
  const T slope0 = (y1 - y0) / dx0;
  const T slope1 = (y2 - y1) / dx1;
  T k0, k1, k2;
  if (periodic) {
    if (y0 != y2) atomicAdd(&A.status->periodic_mismatch, 1ull);
    const T v = (slope0 / dx0 + slope1 / dx1) / (one / dx0 + one / dx1);
    k0 = v; k1 = v; k2 = v;
  } else {
    const T r0 = slope0 * two;
    const T r1 = (slope1 * dx0 + slope0 * dx1) * three - A.w[1] * r0;
    const T r2 = slope1 * two - A.w[2] * r1;
    k2 = r2 / A.midp[2];
    k1 = (r1 - A.up[1] * k2) / A.midp[1];
    k0 = (r0 - A.up[0] * k1) / A.midp[0];
  }
  A.ca[l] = k0 * dx0 - (y1 - y0);
  A.cb[l] = (y1 - y0) - k1 * dx0;
  A.ca[L + l] = k1 * dx1 - (y2 - y1);
  A.cb[L + l] = (y2 - y1) - k2 * dx1;
  if (A.kout) {
    A.kout[l] = k0;
    A.kout[L + l] = k1;
    A.kout[2 * L + l] = k2;
  }
}

// SPLINE_PERIODIC, n >= 4 (:498-565): condensed (n-2) system, k = k1 + k_{n-2} * k2.
template <class T>
__global__ __launch_bounds__(64) void spline_build_periodic_kernel(BuildArgs<T> A) {
  const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= A.lanes) return;
  const uint64_t n = A.n, L = A.lanes, m = n - 2;
  const T three = T(3);
  const T* y = A.data + l;
  T* sa = A.ca + l;
  T* sb = A.cb + l;
  const T dx0 = const_load(A.dx, 0), dxl = const_load(A.dx, n - 2), dxl2 = const_load(A.dx, n - 3);
  const T y0 = y[0], y1 = y[L];
  const T yn1 = y[(n - 1) * L], yn2 = y[(n - 2) * L], yn3 = y[(n - 3) * L];
  if (y0 != yn1) atomicAdd(&A.status->periodic_mismatch, 1ull);
  const T slope0 = (y1 - y0) / dx0;
  const T slope_1 = (yn1 - yn2) / dxl;
  const T slope_2 = (yn2 - yn3) / dxl2;
  const T rhs_first = (slope_1 * dx0 + slope0 * dxl) * three;
  const T rhs_last = (slope_2 * dxl + slope_1 * dxl2) * three;  // row n-2 of the sliced rhs
  // forward sweep over rows 0..m-1 (row 0 special, rows 1..m-1 interior formula)
  T r_prev = rhs_first;
  sa[0] = r_prev;
  T ym = y0, yc = y1, yp = y[2 * L];
  for (uint64_t i = 1; i < m; ++i) {
    const T dxn = const_load(A.dx, i), dxn_1 = const_load(A.dx, i - 1);
    const T rhs = three * (dxn * (yc - ym) / dxn_1 + dxn_1 * (yp - yc) / dxn);
    const T r = rhs - const_load(A.w, i) * r_prev;
    sa[i * L] = r;
    r_prev = r;
    ym = yc;
    yc = yp;
    yp = y[(i + 2) * L];
  }
  // back substitution -> k1 rows 0..m-1 parked in cb
  T k_next = r_prev / const_load(A.midp, m - 1);
  sb[(m - 1) * L] = k_next;
  for (uint64_t i = m - 1; i-- > 0;) {
    const T k = (sa[i * L] - const_load(A.up, i) * k_next) / const_load(A.midp, i);
    sb[i * L] = k;
    k_next = k;
  }
  const T k1_first = sb[0], k1_last = sb[(m - 1) * L];
  const T k_m1 = (rhs_last - k1_first * dxl2 - k1_last * dxl) / A.per_den;
  const T k_0 = k1_first + k_m1 * const_load(A.k2, 0);
  // k[i] = k1[i] + k_m1*k2[i] (i < m), k[m] = k_m1, k[n-1] = k[0]; a/b :354-365
  T k_i = k_0;
  T y_lo = y0;
  for (uint64_t i = 0; i + 1 < n; ++i) {
    T k_r;
    if (i + 1 < m) k_r = sb[(i + 1) * L] + k_m1 * const_load(A.k2, i + 1);
    else if (i + 1 == m) k_r = k_m1;
    else k_r = k_0;
    const T y_hi = y[(i + 1) * L];
    const T dy = y_hi - y_lo;
    const T dxi = const_load(A.dx, i);
    sa[i * L] = k_i * dxi - dy;
    sb[i * L] = dy - k_r * dxi;
    if (A.kout) {
      A.kout[i * L + l] = k_i;
      if (i + 2 == n) A.kout[(i + 1) * L + l] = k_r;
    }
    k_i = k_r;
    y_lo = y_hi;
  }
}

}  // namespace ndi
