// host/ndarray_interp.hpp -- C++17 host-side mirror of the reference's builder / strategy surface
// above the C ABI (include/ndinterp.h).
//
// The reference is Rust (no toolchain in this image), so the compiled host side is written in C++:
// same names, argument meaning and error behaviour as src/interp1d/mod.rs, src/interp2d/mod.rs and the
// strategy modules, so a test written against it reads like the reference's own tests
// (tests/cpp/test_host_mirror.cpp).  Header-only; link with -lndinterp_hip.
//
//   auto interp = Interp1DBuilder<double>::new_(data).x(x).strategy(CubicSpline<double>::new_()).build();
//   Array<double> ys = interp.interp_array(xs);          // one ndi_interp1d_eval call
//
// Rust traits -> abstract classes.  `Interp1DStrategy::interp_into` is the reference's per-query hook
// (strategies/mod.rs:59-64); `interp_array_into` is the defaulted batched hook whose default body is the
// reference's serial loop (interp1d/mod.rs:326-343), overridden by the built-in strategies.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/ndinterp.h"

namespace ndarray_interp {

// ---- errors (src/lib.rs:127-146) ---------------------------------------------------------------
struct BuilderError : std::runtime_error {
  enum Kind { NotEnoughData, Monotonic, ShapeError, ValueError } kind;
  BuilderError(Kind k, const std::string& m) : std::runtime_error(m), kind(k) {}
};
struct InterpolateError : std::runtime_error {
  enum Kind { OutOfBounds } kind = OutOfBounds;
  uint64_t index;  // lowest failing flat query index
  double value;
  int axis;
  InterpolateError(const std::string& m, uint64_t i, double v, int a)
      : std::runtime_error(m), index(i), value(v), axis(a) {}
};
struct Panic : std::logic_error {  // conditions on which the reference panics
  using std::logic_error::logic_error;
};
struct DeviceError : std::runtime_error {  // HIP failure / no device / unsupported: no CPU fallback
  using std::runtime_error::runtime_error;
};

namespace detail {
template <class T> struct DType;
template <> struct DType<float> { static constexpr int id = NDI_F32; };
template <> struct DType<double> { static constexpr int id = NDI_F64; };

inline std::string rust_float(double v) {  // `{x:#?}` of a float
  if (v != v) return "NaN";
  if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
  char buf[64];
  for (int prec = 1; prec <= 17; ++prec) {
    snprintf(buf, sizeof buf, "%.*g", prec, v);
    if (std::strtod(buf, nullptr) == v) break;
  }
  std::string s(buf);
  if (s.find_first_of(".en") == std::string::npos) s += ".0";
  return s;
}
[[noreturn]] inline void throw_builder(int st) {
  const std::string m = ndi_last_error_string();
  switch (st) {
    case NDI_NOT_ENOUGH_DATA: throw BuilderError(BuilderError::NotEnoughData, m);
    case NDI_MONOTONIC: throw BuilderError(BuilderError::Monotonic, m);
    case NDI_SHAPE: throw BuilderError(BuilderError::ShapeError, m);
    case NDI_VALUE: throw BuilderError(BuilderError::ValueError, m);
    default: throw DeviceError("ndinterp_hip status " + std::to_string(st) + ": " + m);
  }
}
[[noreturn]] inline void throw_eval(int st, const ndi_oob_info& info) {
  if (st == NDI_OUT_OF_BOUNDS)  // linear.rs:81-83, cubic_spline.rs:799-801, bilinear.rs:72-79
    throw InterpolateError(std::string(info.axis == 0 ? "x" : "y") + " = " + rust_float(info.value) +
                               " is not in range", info.index, info.value, info.axis);
  if (st == NDI_NAN_QUERY) throw Panic("not implemented: failed to convert NaN to usize");
  throw DeviceError("ndinterp_hip status " + std::to_string(st) + ": " + ndi_last_error_string());
}
inline size_t prod(const std::vector<size_t>& s, size_t from = 0) {
  size_t p = 1;
  for (size_t i = from; i < s.size(); ++i) p *= s[i];
  return p;
}
inline std::string shape_str(const std::vector<size_t>& s) {
  std::string r = "[";
  for (size_t i = 0; i < s.size(); ++i) r += (i ? ", " : "") + std::to_string(s[i]);
  return r + "]";
}
}  // namespace detail

// ---- device selection ---------------------------------------------------------------------------------
// The reference has no notion of a device; this mirror adds one build-side option (SURVEY 5 "config / flags"):
// every built-in strategy builder has `.device(ordinal)`, and builders that are not told use `current_device()`,
// a per-thread default (0 unless `set_current_device` was called) -- so a worker thread that owns device d can
// call `set_current_device(d)` once and build unchanged code.
namespace detail {
inline int& tls_device() {
  static thread_local int d = 0;
  return d;
}
}  // namespace detail
inline int current_device() { return detail::tls_device(); }
inline void set_current_device(int ordinal) {
  if (ordinal < 0 || ordinal >= ndi_device_count())
    throw DeviceError("device ordinal " + std::to_string(ordinal) + " out of range [0, " +
                      std::to_string(ndi_device_count()) + ")");
  detail::tls_device() = ordinal;
}
inline int device_count() { return ndi_device_count(); }

// ---- a minimal owned C-order n-d array (the host "ndarray") --------------------------------------
template <class T>
struct Array {
  std::vector<size_t> shape;
  std::vector<T> data;
  Array() = default;
  Array(std::vector<size_t> s, T fill = T(0)) : shape(std::move(s)), data(detail::prod(shape), fill) {}
  Array(std::vector<size_t> s, std::vector<T> d) : shape(std::move(s)), data(std::move(d)) {
    if (data.size() != detail::prod(shape)) throw Panic("Array: data length does not match shape");
  }
  static Array from_vec(std::vector<T> v) { size_t n = v.size(); return Array({n}, std::move(v)); }
  static Array linspace(T a, T b, size_t n) {
    std::vector<T> v(n);
    const T step = n > 1 ? (b - a) / T(n - 1) : T(0);
    for (size_t i = 0; i < n; ++i) v[i] = a + step * T(i);
    if (n > 1) v[n - 1] = b;
    return from_vec(std::move(v));
  }
  size_t ndim() const { return shape.size(); }
  size_t len() const { return data.size(); }
  T& operator[](size_t i) { return data[i]; }
  const T& operator[](size_t i) const { return data[i]; }
};

// ---- VectorExtensions (src/vector_extensions.rs) -----------------------------------------------------
enum class Monotonic { NotMonotonic, RisingStrict, Rising, FallingStrict, Falling };

namespace detail {
// The state machine of src/vector_extensions.rs:116-198 for element types outside f32 / f64 (the i32 axes of
// tests/interp1d.rs:123-140): the library's ndi_monotonic_prop covers the float types.
template <class T>
Monotonic monotonic_generic(const std::vector<T>& v) {
  if (v.size() <= 1) return Monotonic::NotMonotonic;
  bool strict = true, rising = false, falling = false;
  for (size_t i = 0; i + 1 < v.size(); ++i) {
    if (v[i] < v[i + 1]) rising = true;
    else if (v[i] > v[i + 1]) falling = true;
    else strict = false;
    if (rising && falling) return Monotonic::NotMonotonic;
  }
  if (rising) return strict ? Monotonic::RisingStrict : Monotonic::Rising;
  if (falling) return strict ? Monotonic::FallingStrict : Monotonic::Falling;
  return Monotonic::NotMonotonic;
}
// VectorExtensions::get_lower_index for one query, generic element type (vector_extensions.rs:55-111): the unique
// i with k[i] <= x < k[i+1], clamped to [0, n-2]
template <class T>
size_t lower_index_generic(const std::vector<T>& k, T x) {
  const size_t n = k.size();
  if (x <= k[0]) return 0;
  if (x >= k[n - 1]) return n - 2;
  size_t lo = 0, hi = n - 1;   // invariant k[lo] <= x < k[hi]
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) / 2;
    if (k[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}
// Linear::calc_frac (linear.rs:29-36) with the element type's own arithmetic (integer division truncates)
// Integer T: every intermediate must be representable in T, as in the reference's `T` arithmetic -- a Rust debug
// build panics with "attempt to subtract / multiply / add with overflow" (a release build wraps; values that only
// exist through wrapping are not reproduced, the panic is).  Same behaviour as generic_host.py.
template <class T>
T calc_frac(T x1, T y1, T x2, T y2, T x) {
  if constexpr (std::is_integral_v<T>) {
    T dy, dx, d, p, r;
    if (__builtin_sub_overflow(y2, y1, &dy) || __builtin_sub_overflow(x2, x1, &dx))
      throw Panic("attempt to subtract with overflow");
    if (dx == 0) throw Panic("attempt to divide by zero");
    if constexpr (std::is_signed_v<T>)
      if (dy == std::numeric_limits<T>::min() && dx == T(-1)) throw Panic("attempt to divide with overflow");
    const T m = dy / dx;
    if (__builtin_sub_overflow(x, x1, &d)) throw Panic("attempt to subtract with overflow");
    if (__builtin_mul_overflow(m, d, &p)) throw Panic("attempt to multiply with overflow");
    if (__builtin_add_overflow(p, y1, &r)) throw Panic("attempt to add with overflow");
    return r;
  } else {
    const T m = (y2 - y1) / (x2 - x1);
    return m * (x - x1) + y1;
  }
}
template <class T>
std::string debug_value(T v) {
  if constexpr (std::is_floating_point_v<T>) return rust_float((double)v);
  else return std::to_string(v);
}
}  // namespace detail

template <class T>
Monotonic monotonic_prop(const std::vector<T>& v) {
  if constexpr (!std::is_same_v<T, float> && !std::is_same_v<T, double>) {
    return detail::monotonic_generic(v);
  } else {
    switch (ndi_monotonic_prop(detail::DType<T>::id, v.data(), v.size())) {
      case NDI_MONO_RISING_STRICT: return Monotonic::RisingStrict;
      case NDI_MONO_RISING: return Monotonic::Rising;
      case NDI_MONO_FALLING_STRICT: return Monotonic::FallingStrict;
      case NDI_MONO_FALLING: return Monotonic::Falling;
      default: return Monotonic::NotMonotonic;
    }
  }
}
template <class T>
std::vector<int64_t> get_lower_index(const std::vector<T>& knots, const std::vector<T>& xs, int device = -1) {
  if (device < 0) device = current_device();
  std::vector<int64_t> out(xs.size());
  if constexpr (!std::is_same_v<T, float> && !std::is_same_v<T, double>) {
    for (size_t i = 0; i < xs.size(); ++i) out[i] = (int64_t)detail::lower_index_generic(knots, xs[i]);
  } else {
    int st = ndi_get_lower_index_batch(detail::DType<T>::id, device, knots.data(), knots.size(), xs.data(),
                                       xs.size(), out.data(), NDI_MEM_HOST);
    if (st != NDI_OK) throw DeviceError(ndi_last_error_string());
  }
  return out;
}

// get_lower_index with the knot pyramid resident on the device (ndi_locator_*): no allocation / knot upload per call
template <class T>
class Locator {
  ndi_locator* h_ = nullptr;
 public:
  explicit Locator(const std::vector<T>& knots, int device = -1) {
    if (device < 0) device = current_device();
    int st = ndi_locator_create(detail::DType<T>::id, device, knots.data(), knots.size(), NDI_MEM_HOST, &h_);
    if (st != NDI_OK) throw DeviceError(ndi_last_error_string());
  }
  Locator(const Locator&) = delete;
  Locator& operator=(const Locator&) = delete;
  ~Locator() { ndi_locator_destroy(h_); }
  std::vector<int64_t> get_lower_index(const std::vector<T>& xs) const {
    std::vector<int64_t> out(xs.size());
    int st = ndi_locator_eval(h_, xs.data(), xs.size(), out.data(), NDI_MEM_HOST, nullptr);
    if (st != NDI_OK) throw DeviceError(ndi_last_error_string());
    return out;
  }
};

// =================================================================================================
// 1-D
// =================================================================================================
template <class T> class Interp1D;

template <class T>
struct Interp1DStrategy {  // trait Interp1DStrategy, strategies/mod.rs:42-65
  virtual ~Interp1DStrategy() = default;
  virtual void interp_into(const Interp1D<T>& interpolator, T* target, T x) const = 0;
  // defaulted batched hook: the reference's loop, stops at the first error
  virtual void interp_array_into(const Interp1D<T>& interpolator, const T* xs, size_t nq, T* out,
                                 size_t row_stride) const {
    for (size_t i = 0; i < nq; ++i) {
      try {
        interp_into(interpolator, out + i * row_stride, xs[i]);
      } catch (InterpolateError& e) {
        e.index = i;
        throw;
      }
    }
  }
};

template <class T>
struct Interp1DStrategyBuilder {  // trait Interp1DStrategyBuilder, strategies/mod.rs:12-40
  virtual ~Interp1DStrategyBuilder() = default;
  virtual size_t MINIMUM_DATA_LENGHT() const = 0;  // sic
  virtual std::shared_ptr<Interp1DStrategy<T>> build(const std::vector<T>& x, const Array<T>& data) = 0;
};

// boundary conditions (cubic_spline.rs:153-217)
struct SingleBoundary {
  int kind = NDI_BC_NOT_A_KNOT;
  double value = 0.0;
  static SingleBoundary NotAKnot() { return {NDI_BC_NOT_A_KNOT, 0.0}; }
  static SingleBoundary Natural() { return {NDI_BC_NATURAL, 0.0}; }
  static SingleBoundary Clamped() { return {NDI_BC_CLAMPED, 0.0}; }
  static SingleBoundary FirstDeriv(double v) { return {NDI_BC_FIRST_DERIV, v}; }
  static SingleBoundary SecondDeriv(double v) { return {NDI_BC_SECOND_DERIV, v}; }
};
struct RowBoundary {   // cubic_spline.rs:170-202: the boundary of one data row
  SingleBoundary left, right;
  static RowBoundary NotAKnot() { return {SingleBoundary::NotAKnot(), SingleBoundary::NotAKnot()}; }
  static RowBoundary Natural() { return {SingleBoundary::Natural(), SingleBoundary::Natural()}; }
  static RowBoundary Clamped() { return {SingleBoundary::Clamped(), SingleBoundary::Clamped()}; }
  static RowBoundary Mixed(SingleBoundary l, SingleBoundary r) { return {l, r}; }
};
struct BoundaryCondition {
  bool periodic = false;
  SingleBoundary left, right;
  // BoundaryCondition::Individual (cubic_spline.rs:165-167, 332-347): one RowBoundary per trailing element, shape
  // [1, data.shape[1..]] flattened in C order; empty = one boundary for the whole dataset
  std::vector<RowBoundary> rows;
  std::vector<size_t> rows_shape;
  static BoundaryCondition Individual(std::vector<size_t> shape, std::vector<RowBoundary> r) {
    BoundaryCondition b;
    b.rows = std::move(r);
    b.rows_shape = std::move(shape);
    return b;
  }
  static BoundaryCondition whole(bool per, SingleBoundary l, SingleBoundary r) {
    BoundaryCondition b;
    b.periodic = per; b.left = l; b.right = r;
    return b;
  }
  static BoundaryCondition NotAKnot() { return whole(false, SingleBoundary::NotAKnot(), SingleBoundary::NotAKnot()); }
  static BoundaryCondition Natural() { return whole(false, SingleBoundary::Natural(), SingleBoundary::Natural()); }
  static BoundaryCondition Clamped() { return whole(false, SingleBoundary::Clamped(), SingleBoundary::Clamped()); }
  static BoundaryCondition Periodic() { return whole(true, {}, {}); }
  // RowBoundary::Mixed applied to the whole dataset
  static BoundaryCondition Mixed(SingleBoundary l, SingleBoundary r) { return whole(false, l, r); }
};

namespace detail {
template <class T>
struct Device1D : Interp1DStrategy<T> {  // owns an ndi_interp1d*
  ndi_interp1d* h = nullptr;
  size_t lanes = 1;
  int path = NDI_PATH_AUTO;
  int device = 0;   // the HIP device that holds the tables
  ~Device1D() override { ndi_interp1d_destroy(h); }
  void create(const std::vector<T>& x, const Array<T>& data, int strategy, bool extrapolate, bool periodic,
              ndi_boundary left, ndi_boundary right, int device_, const std::vector<RowBoundary>* rows = nullptr) {
    device = device_;
    ndi_interp1d_desc d{};
    std::vector<int32_t> lk, rk;
    std::vector<double> lv, rv;
    if (rows) {   // BoundaryCondition::Individual -> the four lane_* arrays of the C ABI
      for (const RowBoundary& r : *rows) {
        lk.push_back(r.left.kind); lv.push_back(r.left.value);
        rk.push_back(r.right.kind); rv.push_back(r.right.value);
      }
      d.lane_left_kind = lk.data(); d.lane_left_value = lv.data();
      d.lane_right_kind = rk.data(); d.lane_right_value = rv.data();
    }
    d.dtype = DType<T>::id; d.strategy = strategy; d.extrapolate = extrapolate; d.device = device;
    d.n = data.shape[0]; d.lanes = lanes = prod(data.shape, 1); d.x_len = x.size();
    d.x = x.data(); d.data = data.data.data(); d.memspace = NDI_MEM_HOST;
    d.validate = 0;  // Interp1DBuilder::build validated already (interp1d/mod.rs:449-473)
    d.periodic = periodic; d.left = left; d.right = right;
    int st = ndi_interp1d_create(&d, &h);
    if (st != NDI_OK) throw_builder(st);
  }
  // a replica on `dev` (ndi_interp1d_clone): tables copied device to device, nothing uploaded or solved again
  std::shared_ptr<Device1D<T>> clone(int dev) const {
    auto s = std::make_shared<Device1D<T>>();
    s->lanes = lanes; s->path = path; s->device = dev;
    int st = ndi_interp1d_clone(h, dev, &s->h);
    if (st != NDI_OK) throw_builder(st);
    return s;
  }
  void interp_array_into(const Interp1D<T>&, const T* xs, size_t nq, T* out, size_t row_stride) const override {
    ndi_eval_opts o{};
    o.q_memspace = NDI_MEM_HOST; o.out_memspace = NDI_MEM_HOST; o.path = path;
    ndi_oob_info info{};
    int st = ndi_interp1d_eval(h, xs, nq, out, row_stride, &o, &info);
    if (st != NDI_OK) throw_eval(st, info);
  }
  void interp_into(const Interp1D<T>& i, T* target, T x) const override { interp_array_into(i, &x, 1, target, lanes); }
};

// Linear::interp_into (linear.rs:73-98) for element types the device path does not cover (integers ...): the
// reference's generic per-query body; batches take the trait's default loop.
template <class T>
struct HostLinear : Interp1DStrategy<T> {
  bool extrapolate = false;
  void interp_into(const Interp1D<T>& ip, T* target, T x) const override;
};
}  // namespace detail

// One chunk of a ring evaluation as the consumer sees it (ndi_ring_chunk): rows [q_begin, q_begin + q_count) of
// the batch, T[q_count][row_stride] in device memory at `out`, produced on `stream`.
using RingChunk = ndi_ring_chunk;

// Linear (src/interp1d/strategies/linear.rs): builder and finished strategy in one, as in the reference
template <class T>
class Linear : public Interp1DStrategyBuilder<T> {
  bool extrapolate_ = false;
  int device_ = -1;   // -1: current_device() at build time
 public:
  static Linear new_() { return Linear(); }
  Linear extrapolate(bool e) && { extrapolate_ = e; return std::move(*this); }
  Linear& extrapolate(bool e) & { extrapolate_ = e; return *this; }
  Linear device(int ordinal) && { device_ = ordinal; return std::move(*this); }
  size_t MINIMUM_DATA_LENGHT() const override { return 2; }  // linear.rs:52
  std::shared_ptr<Interp1DStrategy<T>> build(const std::vector<T>& x, const Array<T>& data) override {
    if constexpr (!std::is_same_v<T, float> && !std::is_same_v<T, double>) {
      auto s = std::make_shared<detail::HostLinear<T>>();   // integers: the reference's generic per-query path
      s->extrapolate = extrapolate_;
      return s;
    } else {
      auto s = std::make_shared<detail::Device1D<T>>();
      s->create(x, data, NDI_LINEAR, extrapolate_, false, {0, 0.0}, {0, 0.0}, device_ >= 0 ? device_ : current_device());
      return s;
    }
  }
};

template <class T>
class CubicSpline : public Interp1DStrategyBuilder<T> {  // cubic_spline.rs:85-88, 723-771
  bool extrapolate_ = false;
  BoundaryCondition boundary_ = BoundaryCondition::NotAKnot();  // default :724-729
  int device_ = -1;   // -1: current_device() at build time
 public:
  static CubicSpline new_() { return CubicSpline(); }
  CubicSpline extrapolate(bool e) && { extrapolate_ = e; return std::move(*this); }
  CubicSpline device(int ordinal) && { device_ = ordinal; return std::move(*this); }
  CubicSpline boundary(BoundaryCondition b) && { boundary_ = b; return std::move(*this); }
  size_t MINIMUM_DATA_LENGHT() const override { return 3; }  // cubic_spline.rs:751
  std::shared_ptr<Interp1DStrategy<T>> build(const std::vector<T>& x, const Array<T>& data) override {
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, double>,
                  "CubicSpline needs a float element type (the reference's trait bounds: Pow / Euclid on T)");
    auto s = std::make_shared<detail::Device1D<T>>();
    const int dev = device_ >= 0 ? device_ : current_device();
    if (!boundary_.rows.empty() || !boundary_.rows_shape.empty()) {   // Individual: shape [1, data.shape[1..]] (:332-340)
      std::vector<size_t> expect{1};
      expect.insert(expect.end(), data.shape.begin() + 1, data.shape.end());
      if (boundary_.rows_shape != expect || boundary_.rows.size() != detail::prod(expect))
        throw BuilderError(BuilderError::ShapeError, "Boundary conditions array has wrong shape. Expected: " +
                                                         detail::shape_str(expect) + ", got: " +
                                                         detail::shape_str(boundary_.rows_shape));
      s->create(x, data, NDI_CUBIC_SPLINE, extrapolate_, false, {0, 0.0}, {0, 0.0}, dev, &boundary_.rows);
      return s;
    }
    s->create(x, data, NDI_CUBIC_SPLINE, extrapolate_, boundary_.periodic,
              {boundary_.left.kind, boundary_.left.value}, {boundary_.right.kind, boundary_.right.value}, dev);
    return s;
  }
};

template <class T>
class Interp1D {  // interp1d/mod.rs:39-51
 public:
  std::vector<T> x;
  Array<T> data;
  std::shared_ptr<Interp1DStrategy<T>> strategy;

  static Interp1D new_unchecked(std::vector<T> x, Array<T> data, std::shared_ptr<Interp1DStrategy<T>> s) {
    Interp1D i; i.x = std::move(x); i.data = std::move(data); i.strategy = std::move(s); return i;
  }
  size_t lanes() const { return detail::prod(data.shape, 1); }
  std::vector<size_t> lanes_shape() const { return {data.shape.begin() + 1, data.shape.end()}; }

  // A replica of this interpolator whose tables live on `device` (copied device to device): what
  // interp_array_sharded takes, one per device.
  Interp1D replicate(int device) const {
    auto dev = std::dynamic_pointer_cast<detail::Device1D<T>>(strategy);
    if (!dev) throw Panic("replicate needs a built-in device strategy (f32 / f64 data)");
    return new_unchecked(x, data, dev->clone(device));
  }

  std::pair<T, const T*> index_point(size_t index) const { return {x.at(index), data.data.data() + index * lanes()}; }
  size_t get_index_left_of(T v) const {
    auto r = get_lower_index<T>(x, {v});
    if (r[0] < 0) throw Panic("not implemented: failed to convert NaN to usize");
    return (size_t)r[0];
  }
  bool is_in_range(T v) const { return x.front() <= v && v <= x.back(); }

  // interp_array (interp1d/mod.rs:197-211) for outputs larger than device memory: chunks of `chunk_queries` rows
  // through a library-owned device ring of `n_slots` slots (ndi_interp1d_eval_ring); `consume` sees every chunk
  // once, in order, after its kernels are enqueued on chunk.stream (return a hipEvent_t recorded on another
  // stream if the chunk is drained there, else nullptr).  Needs a built-in device strategy.
  void interp_array_ring(const Array<T>& xs, size_t chunk_queries, unsigned n_slots,
                         const std::function<void*(const RingChunk&)>& consume) const {
    auto dev = std::dynamic_pointer_cast<detail::Device1D<T>>(strategy);
    if (!dev) throw Panic("interp_array_ring needs a built-in device strategy (f32 / f64 data)");
    ndi_ring_desc ring{};
    ring.n_slots = n_slots;
    ring.chunk_queries = chunk_queries;
    ndi_eval_opts o{};
    o.q_memspace = NDI_MEM_HOST; o.out_memspace = NDI_MEM_DEVICE; o.path = dev->path;
    ndi_oob_info info{};
    // an exception of the consumer must not unwind through the C ABI: keep the first one, stop consuming, rethrow
    struct Ctx { const std::function<void*(const RingChunk&)>* fn; std::exception_ptr err; } ctx{&consume, nullptr};
    auto tramp = [](void* user, const ndi_ring_chunk* c) -> void* {
      Ctx* cx = static_cast<Ctx*>(user);
      if (cx->err) return nullptr;
      try { return (*cx->fn)(*c); } catch (...) { cx->err = std::current_exception(); return nullptr; }
    };
    int st = ndi_interp1d_eval_ring(dev->h, xs.data.data(), xs.len(), &ring, tramp, &ctx, &o, &info);
    if (ctx.err) std::rethrow_exception(ctx.err);
    if (st != NDI_OK) detail::throw_eval(st, info);
  }

  T interp_scalar(T v) const {  // :108-114
    if (data.ndim() != 1) throw Panic("interp_scalar needs 1-D data");
    T out = T(0);
    strategy->interp_into(*this, &out, v);
    return out;
  }
  Array<T> interp(T v) const {  // :150-156
    Array<T> target(lanes_shape());
    strategy->interp_into(*this, target.data.data(), v);
    return target;
  }
  void interp_into(T v, Array<T>& buffer) const {  // :169-175
    if (buffer.shape != lanes_shape())
      throw Panic("incompatible shapes expected: " + detail::shape_str(lanes_shape()) + ", got: " +
                  detail::shape_str(buffer.shape));
    strategy->interp_into(*this, buffer.data.data(), v);
  }
  std::vector<size_t> get_buffer_shape(const std::vector<size_t>& dq) const {  // :346-354
    std::vector<size_t> s = dq;
    s.insert(s.end(), data.shape.begin() + 1, data.shape.end());
    return s;
  }
  Array<T> interp_array(const Array<T>& xs) const {  // :197-211
    Array<T> ys(get_buffer_shape(xs.shape));
    interp_array_into(xs, ys);
    return ys;
  }
  void interp_array_into(const Array<T>& xs, Array<T>& buffer) const {  // :272-324, any query rank
    const auto expect = get_buffer_shape(xs.shape);
    if (buffer.shape != expect)
      throw Panic("incompatible shapes expected: " + detail::shape_str(expect) + ", got: " +
                  detail::shape_str(buffer.shape));
    strategy->interp_array_into(*this, xs.data.data(), xs.len(), buffer.data.data(), lanes());
  }
};

template <class T>
void detail::HostLinear<T>::interp_into(const Interp1D<T>& ip, T* target, T x) const {   // linear.rs:73-98
  if (!extrapolate && !ip.is_in_range(x))
    throw InterpolateError("x = " + detail::debug_value(x) + " is not in range", 0, (double)x, 0);
  const size_t i = detail::lower_index_generic(ip.x, x);
  const size_t L = ip.lanes();
  const T* y1 = ip.data.data.data() + i * L;
  const T* y2 = y1 + L;
  for (size_t l = 0; l < L; ++l) target[l] = detail::calc_frac(ip.x[i], y1[l], ip.x[i + 1], y2[l], x);
}

template <class T>
class Interp1DBuilder {  // interp1d/mod.rs:60-70, 389-477
  Array<T> data_;
  std::vector<T> x_;
  bool has_x_ = false;
  std::shared_ptr<Interp1DStrategyBuilder<T>> strategy_;
 public:
  static Interp1DBuilder new_(Array<T> data) {
    Interp1DBuilder b; b.data_ = std::move(data); b.strategy_ = std::make_shared<Linear<T>>(); return b;  // :408
  }
  Interp1DBuilder x(std::vector<T> x) && { x_ = std::move(x); has_x_ = true; return std::move(*this); }
  template <class S> Interp1DBuilder strategy(S s) && { strategy_ = std::make_shared<S>(std::move(s)); return std::move(*this); }

  Interp1D<T> build() && {  // :443-476, check order preserved
    if (data_.ndim() < 1) throw BuilderError(BuilderError::ShapeError, "data dimension is 0, needs to be at least 1");
    const size_t n = data_.shape[0];
    if (!has_x_) { x_.resize(n); for (size_t i = 0; i < n; ++i) x_[i] = T(i); }  // :402-406
    const size_t need = strategy_->MINIMUM_DATA_LENGHT();
    if (n < need)
      throw BuilderError(BuilderError::NotEnoughData,
                         "The chosen Interpolation strategy needs at least " + std::to_string(need) + " data points");
    if (monotonic_prop(x_) != Monotonic::RisingStrict)
      throw BuilderError(BuilderError::Monotonic, "Values in the x axis need to be strictly monotonic rising");
    if (x_.size() != n)
      throw BuilderError(BuilderError::ShapeError, "Lengths of x and data axis need to match. Got x: " +
                                                       std::to_string(x_.size()) + ", data: " + std::to_string(n));
    auto finished = strategy_->build(x_, data_);
    return Interp1D<T>::new_unchecked(std::move(x_), std::move(data_), std::move(finished));
  }
};

// =================================================================================================
// 2-D
// =================================================================================================
template <class T> class Interp2D;

template <class T>
struct Interp2DStrategy {  // src/interp2d/strategies/mod.rs:46-73
  virtual ~Interp2DStrategy() = default;
  virtual void interp_into(const Interp2D<T>&, T* target, T x, T y) const = 0;
  virtual void interp_array_into(const Interp2D<T>& ip, const T* xs, const T* ys, size_t nq, T* out,
                                 size_t row_stride) const {
    for (size_t i = 0; i < nq; ++i) {
      try {
        interp_into(ip, out + i * row_stride, xs[i], ys[i]);
      } catch (InterpolateError& e) {
        e.index = i;
        throw;
      }
    }
  }
};
template <class T>
struct Interp2DStrategyBuilder {  // src/interp2d/strategies/mod.rs:14-44
  virtual ~Interp2DStrategyBuilder() = default;
  virtual size_t MINIMUM_DATA_LENGHT() const = 0;
  virtual std::shared_ptr<Interp2DStrategy<T>> build(const std::vector<T>& x, const std::vector<T>& y,
                                                     const Array<T>& data) = 0;
};

namespace detail {
template <class T>
struct Device2D : Interp2DStrategy<T> {
  ndi_interp2d* h = nullptr;
  size_t lanes = 1;
  int device = 0;
  ~Device2D() override { ndi_interp2d_destroy(h); }
  std::shared_ptr<Device2D<T>> clone(int dev) const {   // ndi_interp2d_clone
    auto s = std::make_shared<Device2D<T>>();
    s->lanes = lanes; s->device = dev;
    int st = ndi_interp2d_clone(h, dev, &s->h);
    if (st != NDI_OK) throw_builder(st);
    return s;
  }
  void interp_array_into(const Interp2D<T>&, const T* xs, const T* ys, size_t nq, T* out,
                         size_t row_stride) const override {
    ndi_eval_opts o{};
    o.q_memspace = NDI_MEM_HOST; o.out_memspace = NDI_MEM_HOST;
    ndi_oob_info info{};
    int st = ndi_interp2d_eval(h, xs, ys, nq, out, row_stride, &o, &info);
    if (st != NDI_OK) throw_eval(st, info);
  }
  void interp_into(const Interp2D<T>& i, T* target, T x, T y) const override {
    interp_array_into(i, &x, &y, 1, target, lanes);
  }
};
// Bilinear::interp_into (bilinear.rs:64-99) for element types the device path does not cover (integers ...)
template <class T>
struct HostBilinear : Interp2DStrategy<T> {
  bool extrapolate = false;
  void interp_into(const Interp2D<T>& ip, T* target, T x, T y) const override;
};
}  // namespace detail

template <class T>
class Bilinear : public Interp2DStrategyBuilder<T> {  // src/interp2d/strategies/bilinear.rs
  bool extrapolate_ = false;
  int device_ = -1;   // -1: current_device() at build time
 public:
  static Bilinear new_() { return Bilinear(); }
  Bilinear extrapolate(bool yes) && { extrapolate_ = yes; return std::move(*this); }
  Bilinear device(int ordinal) && { device_ = ordinal; return std::move(*this); }
  size_t MINIMUM_DATA_LENGHT() const override { return 2; }  // bilinear.rs:41
  std::shared_ptr<Interp2DStrategy<T>> build(const std::vector<T>& x, const std::vector<T>& y,
                                             const Array<T>& data) override {
    if constexpr (!std::is_same_v<T, float> && !std::is_same_v<T, double>) {
      auto s = std::make_shared<detail::HostBilinear<T>>();   // integers: the reference's generic per-query path
      s->extrapolate = extrapolate_;
      return s;
    } else {
      auto s = std::make_shared<detail::Device2D<T>>();
      ndi_interp2d_desc d{};
      s->device = device_ >= 0 ? device_ : current_device();
      d.dtype = detail::DType<T>::id; d.extrapolate = extrapolate_; d.device = s->device; d.memspace = NDI_MEM_HOST;
      d.nx = data.shape[0]; d.ny = data.shape[1]; d.lanes = s->lanes = detail::prod(data.shape, 2);
      d.x_len = x.size(); d.y_len = y.size(); d.x = x.data(); d.y = y.data(); d.data = data.data.data();
      d.validate = 0;
      int st = ndi_interp2d_create(&d, &s->h);
      if (st != NDI_OK) detail::throw_builder(st);
      return s;
    }
  }
};

template <class T>
class Interp2D {  // interp2d/mod.rs:36-48
 public:
  std::vector<T> x, y;
  Array<T> data;
  std::shared_ptr<Interp2DStrategy<T>> strategy;
  size_t lanes() const { return detail::prod(data.shape, 2); }
  std::vector<size_t> lanes_shape() const { return {data.shape.begin() + 2, data.shape.end()}; }
  Interp2D replicate(int device) const {   // see Interp1D::replicate
    auto dev = std::dynamic_pointer_cast<detail::Device2D<T>>(strategy);
    if (!dev) throw Panic("replicate needs a built-in device strategy (f32 / f64 data)");
    Interp2D r;
    r.x = x; r.y = y; r.data = data; r.strategy = dev->clone(device);
    return r;
  }
  bool is_in_x_range(T v) const { return x.front() <= v && v <= x.back(); }
  bool is_in_y_range(T v) const { return y.front() <= v && v <= y.back(); }
  T interp_scalar(T xv, T yv) const {
    if (data.ndim() != 2) throw Panic("interp_scalar needs 2-D data");
    T out = T(0);
    strategy->interp_into(*this, &out, xv, yv);
    return out;
  }
  Array<T> interp(T xv, T yv) const {
    Array<T> t(lanes_shape());
    strategy->interp_into(*this, t.data.data(), xv, yv);
    return t;
  }
  void interp_into(T xv, T yv, Array<T>& buffer) const {  // :150-167; panics on a wrong buffer shape
    if (buffer.shape != lanes_shape())
      throw Panic("incompatible shapes expected: " + detail::shape_str(lanes_shape()) + ", got: " +
                  detail::shape_str(buffer.shape));
    strategy->interp_into(*this, buffer.data.data(), xv, yv);
  }
  // (x knot, y knot, pointer to the lanes of the grid point) -- interp2d/mod.rs:348-364
  struct Point { T x, y; const T* data; };
  Point index_point(size_t x_idx, size_t y_idx) const {
    return {x.at(x_idx), y.at(y_idx), data.data.data() + (x_idx * data.shape[1] + y_idx) * lanes()};
  }
  std::pair<size_t, size_t> get_index_left_of(T xv, T yv) const {  // :370-372
    auto one = [](const std::vector<T>& k, T v) {
      auto r = get_lower_index<T>(k, {v});
      if (r[0] < 0) throw Panic("not implemented: failed to convert NaN to usize");
      return (size_t)r[0];
    };
    return {one(x, xv), one(y, yv)};
  }
  std::vector<size_t> get_buffer_shape(const std::vector<size_t>& dq) const {  // :310-321
    std::vector<size_t> s = dq;
    s.insert(s.end(), data.shape.begin() + 2, data.shape.end());
    return s;
  }
  Array<T> interp_array(const Array<T>& xs, const Array<T>& ys) const {  // :175-196
    if (xs.shape != ys.shape) throw Panic("`xs.shape()` and `ys.shape()` do not match");
    Array<T> zs(get_buffer_shape(xs.shape));
    interp_array_into(xs, ys, zs);
    return zs;
  }
  void interp_array_into(const Array<T>& xs, const Array<T>& ys, Array<T>& buffer) const {  // :215-285
    if (xs.shape != ys.shape) throw Panic("`xs.shape()` and `ys.shape()` do not match");
    const auto expect = get_buffer_shape(xs.shape);
    if (buffer.shape != expect)
      throw Panic("incompatible shapes expected: " + detail::shape_str(expect) + ", got: " +
                  detail::shape_str(buffer.shape));
    strategy->interp_array_into(*this, xs.data.data(), ys.data.data(), xs.len(), buffer.data.data(), lanes());
  }
};

template <class T>
void detail::HostBilinear<T>::interp_into(const Interp2D<T>& ip, T* target, T x, T y) const {   // bilinear.rs:64-99
  if (!extrapolate && !ip.is_in_x_range(x))    // x before y (:71-80)
    throw InterpolateError("x = " + detail::debug_value(x) + " is not in range", 0, (double)x, 0);
  if (!extrapolate && !ip.is_in_y_range(y))
    throw InterpolateError("y = " + detail::debug_value(y) + " is not in range", 0, (double)y, 1);
  const size_t xi = detail::lower_index_generic(ip.x, x), yi = detail::lower_index_generic(ip.y, y);
  const size_t L = ip.lanes(), ny = ip.data.shape[1];
  const T* g = ip.data.data.data();
  const T x1 = ip.x[xi], x2 = ip.x[xi + 1], y1 = ip.y[yi], y2 = ip.y[yi + 1];
  for (size_t l = 0; l < L; ++l) {
    const T z11 = g[(xi * ny + yi) * L + l], z12 = g[(xi * ny + yi + 1) * L + l];
    const T z21 = g[((xi + 1) * ny + yi) * L + l], z22 = g[((xi + 1) * ny + yi + 1) * L + l];
    const T z1 = detail::calc_frac(x1, z11, x2, z21, x);   // :88-97
    const T z2 = detail::calc_frac(x1, z12, x2, z22, x);
    target[l] = detail::calc_frac(y1, z1, y2, z2, y);
  }
}

template <class T>
class Interp2DBuilder {  // interp2d/mod.rs:52-64, 382-519
  Array<T> data_;
  std::vector<T> x_, y_;
  bool has_x_ = false, has_y_ = false;
  std::shared_ptr<Interp2DStrategyBuilder<T>> strategy_;
 public:
  static Interp2DBuilder new_(Array<T> data) {
    Interp2DBuilder b; b.data_ = std::move(data); b.strategy_ = std::make_shared<Bilinear<T>>(); return b;
  }
  Interp2DBuilder x(std::vector<T> v) && { x_ = std::move(v); has_x_ = true; return std::move(*this); }
  Interp2DBuilder y(std::vector<T> v) && { y_ = std::move(v); has_y_ = true; return std::move(*this); }
  template <class S> Interp2DBuilder strategy(S s) && { strategy_ = std::make_shared<S>(std::move(s)); return std::move(*this); }
  Interp2D<T> build() && {  // :468-518, check order preserved
    if (data_.ndim() < 2) throw BuilderError(BuilderError::ShapeError, "data dimension needs to be at least 2");
    const size_t nx = data_.shape[0], ny = data_.shape[1], need = strategy_->MINIMUM_DATA_LENGHT();
    auto ne = [&](int dim, size_t have) {
      return BuilderError(BuilderError::NotEnoughData,
                          "The " + std::to_string(dim) + "-dimension has not enough data for the chosen interpolation "
                          "strategy. Provided: " + std::to_string(have) + ", Reqired: " + std::to_string(need));
    };
    if (nx < need) throw ne(0, nx);
    if (ny < need) throw ne(1, ny);
    if (!has_x_) { x_.resize(nx); for (size_t i = 0; i < nx; ++i) x_[i] = T(i); }
    if (!has_y_) { y_.resize(ny); for (size_t i = 0; i < ny; ++i) y_[i] = T(i); }
    if (x_.size() != nx)
      throw BuilderError(BuilderError::ShapeError, "Lenghts of x-axis and data-0-axis need to match. Got x: " +
                                                       std::to_string(x_.size()) + ", data-0: " + std::to_string(nx));
    if (y_.size() != ny)
      throw BuilderError(BuilderError::ShapeError, "Lenghts of y-axis and data-1-axis need to match. Got y: " +
                                                       std::to_string(y_.size()) + ", data-1: " + std::to_string(ny));
    if (monotonic_prop(x_) != Monotonic::RisingStrict)
      throw BuilderError(BuilderError::Monotonic, "The x-axis needs to be strictly monotonic rising");
    if (monotonic_prop(y_) != Monotonic::RisingStrict)
      throw BuilderError(BuilderError::Monotonic, "The y-axis needs to be strictly monotonic rising");
    Interp2D<T> ip;
    ip.strategy = strategy_->build(x_, y_, data_);
    ip.x = std::move(x_); ip.y = std::move(y_); ip.data = std::move(data_);
    return ip;
  }
};

// =================================================================================================
// several devices, one call (ndi_interp{1,2}d_eval_sharded / _eval_ring_sharded)
// =================================================================================================
// `replicas`: interpolators built from the same arrays, one per device (`.device(d)` on the strategy builder).
// The flattened query array is split into contiguous blocks (ndi_shard_bounds), every block is evaluated by its own
// host thread inside the library on its replica's device, and the result is what the reference's serial loop over
// the whole batch gives (interp1d/mod.rs:326-343): on Err the exception names the global flat index, rows before
// it are written, later rows are untouched.  The reference's multi-worker shape is benches/bench_interp1d.rs:49-79.
inline std::pair<size_t, size_t> shard_bounds(size_t nq, unsigned shard, unsigned n_shards) {
  uint64_t lo = 0, hi = 0;
  ndi_shard_bounds(nq, shard, n_shards, &lo, &hi);
  return {(size_t)lo, (size_t)hi};
}

template <class T>
void interp_array_into_sharded(const std::vector<const Interp1D<T>*>& replicas, const Array<T>& xs, Array<T>& buffer) {
  if (replicas.empty()) throw Panic("interp_array_sharded needs at least one replica");
  const auto expect = replicas[0]->get_buffer_shape(xs.shape);
  if (buffer.shape != expect)
    throw Panic("incompatible shapes expected: " + detail::shape_str(expect) + ", got: " + detail::shape_str(buffer.shape));
  const size_t n = replicas.size(), lanes = replicas[0]->lanes(), nq = xs.len();
  std::vector<const ndi_interp1d*> hs(n);
  std::vector<ndi_shard_io> io(n);
  for (size_t i = 0; i < n; ++i) {
    auto dev = std::dynamic_pointer_cast<detail::Device1D<T>>(replicas[i]->strategy);
    if (!dev) throw Panic("interp_array_sharded needs built-in device strategies (f32 / f64 data)");
    hs[i] = dev->h;
    io[i] = ndi_shard_io{nullptr, nullptr, buffer.data.data() + shard_bounds(nq, (unsigned)i, (unsigned)n).first * lanes, nullptr};
  }
  ndi_eval_opts o{};
  o.q_memspace = NDI_MEM_HOST; o.out_memspace = NDI_MEM_HOST;
  o.path = std::dynamic_pointer_cast<detail::Device1D<T>>(replicas[0]->strategy)->path;
  ndi_oob_info info{};
  int st = ndi_interp1d_eval_sharded(hs.data(), (uint32_t)n, xs.data.data(), nq, io.data(), lanes, &o, &info);
  if (st != NDI_OK) detail::throw_eval(st, info);
}
template <class T>
Array<T> interp_array_sharded(const std::vector<const Interp1D<T>*>& replicas, const Array<T>& xs) {
  if (replicas.empty()) throw Panic("interp_array_sharded needs at least one replica");
  Array<T> ys(replicas[0]->get_buffer_shape(xs.shape));
  interp_array_into_sharded(replicas, xs, ys);
  return ys;
}

// The ring evaluation over several devices: every replica streams its block through its own library-owned ring
// of `n_slots` slots; `consume` is called concurrently from the shards' host threads (chunk.shard names the shard,
// chunk.q_begin is the global flat index).
template <class T>
void interp_array_ring_sharded(const std::vector<const Interp1D<T>*>& replicas, const Array<T>& xs,
                               size_t chunk_queries, unsigned n_slots,
                               const std::function<void*(const RingChunk&)>& consume) {
  const size_t n = replicas.size();
  if (!n) throw Panic("interp_array_ring_sharded needs at least one replica");
  std::vector<const ndi_interp1d*> hs(n);
  std::vector<ndi_ring_desc> rings(n);
  for (size_t i = 0; i < n; ++i) {
    auto dev = std::dynamic_pointer_cast<detail::Device1D<T>>(replicas[i]->strategy);
    if (!dev) throw Panic("interp_array_ring_sharded needs built-in device strategies (f32 / f64 data)");
    hs[i] = dev->h;
    rings[i] = ndi_ring_desc{};
    rings[i].n_slots = n_slots;
    rings[i].chunk_queries = chunk_queries;
  }
  ndi_eval_opts o{};
  o.q_memspace = NDI_MEM_HOST; o.out_memspace = NDI_MEM_DEVICE;
  o.path = std::dynamic_pointer_cast<detail::Device1D<T>>(replicas[0]->strategy)->path;
  ndi_oob_info info{};
  struct Ctx { const std::function<void*(const RingChunk&)>* fn; std::mutex mu; std::exception_ptr err; } ctx{&consume, {}, nullptr};
  auto tramp = [](void* user, const ndi_ring_chunk* c) -> void* {
    Ctx* cx = static_cast<Ctx*>(user);
    {
      std::lock_guard<std::mutex> g(cx->mu);
      if (cx->err) return nullptr;
    }
    try { return (*cx->fn)(*c); } catch (...) {
      std::lock_guard<std::mutex> g(cx->mu);
      if (!cx->err) cx->err = std::current_exception();
      return nullptr;
    }
  };
  int st = ndi_interp1d_eval_ring_sharded(hs.data(), (uint32_t)n, xs.data.data(), xs.len(), nullptr, rings.data(), tramp,
                                          &ctx, &o, &info);
  if (ctx.err) std::rethrow_exception(ctx.err);
  if (st != NDI_OK) detail::throw_eval(st, info);
}

template <class T>
void interp_array_into_sharded(const std::vector<const Interp2D<T>*>& replicas, const Array<T>& xs, const Array<T>& ys,
                               Array<T>& buffer) {
  if (replicas.empty()) throw Panic("interp_array_sharded needs at least one replica");
  if (xs.shape != ys.shape) throw Panic("`xs.shape()` and `ys.shape()` do not match");
  const auto expect = replicas[0]->get_buffer_shape(xs.shape);
  if (buffer.shape != expect)
    throw Panic("incompatible shapes expected: " + detail::shape_str(expect) + ", got: " + detail::shape_str(buffer.shape));
  const size_t n = replicas.size(), lanes = replicas[0]->lanes(), nq = xs.len();
  std::vector<const ndi_interp2d*> hs(n);
  std::vector<ndi_shard_io> io(n);
  for (size_t i = 0; i < n; ++i) {
    auto dev = std::dynamic_pointer_cast<detail::Device2D<T>>(replicas[i]->strategy);
    if (!dev) throw Panic("interp_array_sharded needs built-in device strategies (f32 / f64 data)");
    hs[i] = dev->h;
    io[i] = ndi_shard_io{nullptr, nullptr, buffer.data.data() + shard_bounds(nq, (unsigned)i, (unsigned)n).first * lanes, nullptr};
  }
  ndi_eval_opts o{};
  o.q_memspace = NDI_MEM_HOST; o.out_memspace = NDI_MEM_HOST;
  ndi_oob_info info{};
  int st = ndi_interp2d_eval_sharded(hs.data(), (uint32_t)n, xs.data.data(), ys.data.data(), nq, io.data(), lanes, &o, &info);
  if (st != NDI_OK) detail::throw_eval(st, info);
}
template <class T>
Array<T> interp_array_sharded(const std::vector<const Interp2D<T>*>& replicas, const Array<T>& xs, const Array<T>& ys) {
  if (replicas.empty()) throw Panic("interp_array_sharded needs at least one replica");
  Array<T> zs(replicas[0]->get_buffer_shape(xs.shape));
  interp_array_into_sharded(replicas, xs, ys, zs);
  return zs;
}

}  // namespace ndarray_interp
