// tests/cpp/test_host_mirror.cpp -- the C++ host mirror (ndarray-interp_amd/host/ndarray_interp.hpp) driven the
// way the reference's own tests drive the crate (tests/interp1d.rs, tests/interp2d.rs,
// tests/cubic_spline_strat.rs, examples/custom_strategy.rs).  `--host-only` runs the cases that need no GPU
// (builder validation); without it the device cases run too.  Exit code = number of failed checks.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>

#include "../../ndarray-interp_amd/host/ndarray_interp.hpp"

using namespace ndarray_interp;
static int failures = 0;
// the two HIP runtime calls the ring consumer below needs (plain C prototypes; the mirror itself needs no HIP header)
extern "C" int hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                                int kind, void* stream);
extern "C" int hipStreamSynchronize(void* stream);

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) { ++failures; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)
template <class E, class F>
static bool throws(F&& f) {
  try { f(); } catch (const E&) { return true; } catch (...) { return false; }
  return false;
}
template <class E, class F>
static bool throws(F&& f, int kind) {
  try { f(); } catch (const E& e) { return (int)e.kind == kind; } catch (...) { return false; }
  return false;
}
static Array<double> arr(std::vector<double> v) { return Array<double>::from_vec(std::move(v)); }

static void host_only() {
  // tests/interp1d.rs:123-140 (builder errors) and the check order of interp1d/mod.rs:454-471
  CHECK(throws<BuilderError>([] { Interp1DBuilder<double>::new_(arr({1})).build(); }, BuilderError::NotEnoughData));
  CHECK(throws<BuilderError>([] { Interp1DBuilder<double>::new_(arr({1, 2})).x({1, 2, 3}).build(); }, BuilderError::ShapeError));
  CHECK(throws<BuilderError>([] { Interp1DBuilder<double>::new_(arr({1, 2, 3})).x({1, 2, 2}).build(); }, BuilderError::Monotonic));
  CHECK(throws<BuilderError>([] { Interp1DBuilder<double>::new_(arr({1, 2})).strategy(CubicSpline<double>::new_()).build(); },
                             BuilderError::NotEnoughData));  // tests/cubic_spline_strat.rs:30-35
  CHECK(throws<BuilderError>([] { Interp1DBuilder<double>::new_(arr({1})).x({2, 1}).build(); }, BuilderError::NotEnoughData));
  // tests/interp2d.rs:280-329
  auto g = [](size_t a, size_t b) { return Array<double>({a, b}, 1.0); };
  CHECK(throws<BuilderError>([&] { Interp2DBuilder<double>::new_(g(1, 1)).build(); }, BuilderError::NotEnoughData));
  CHECK(throws<BuilderError>([&] { Interp2DBuilder<double>::new_(g(1, 2)).build(); }, BuilderError::NotEnoughData));
  CHECK(throws<BuilderError>([&] { Interp2DBuilder<double>::new_(g(2, 1)).build(); }, BuilderError::NotEnoughData));
  CHECK(throws<BuilderError>([&] { Interp2DBuilder<double>::new_(g(2, 2)).x({1}).build(); }, BuilderError::ShapeError));
  CHECK(throws<BuilderError>([&] { Interp2DBuilder<double>::new_(g(2, 2)).y({1, 2, 3}).build(); }, BuilderError::ShapeError));
  CHECK(throws<BuilderError>([&] { Interp2DBuilder<double>::new_(g(2, 2)).x({2, 2}).build(); }, BuilderError::Monotonic));
  CHECK(throws<BuilderError>([&] { Interp2DBuilder<double>::new_(g(2, 2)).y({2, 2}).build(); }, BuilderError::Monotonic));
  CHECK(monotonic_prop<double>({1.1, 2.0, 3.123, 4.5}) == Monotonic::RisingStrict);  // vector_extensions.rs:318-346
  CHECK(monotonic_prop<double>({1.1, 2.0, 3.123, 3.123, 4.5}) == Monotonic::Rising);
  CHECK(monotonic_prop<float>({5.8f, 4.1f, 3.1f, 3.1f, 2.0f}) == Monotonic::Falling);
  CHECK(monotonic_prop<double>({1, 1, 1}) == Monotonic::NotMonotonic);
  {  // integer overflow = the reference's debug-build panic (tests/golden/reference_integer_vectors.json,
     // derived_integer_overflow): u32 descending values, i32 difference / product out of range
    auto ud = Interp1DBuilder<unsigned>::new_(Array<unsigned>::from_vec({10u, 4u})).x({0u, 2u}).build();
    CHECK(throws<Panic>([&] { ud.interp_scalar(1u); }));
    auto uu = Interp1DBuilder<unsigned>::new_(Array<unsigned>::from_vec({4u, 10u})).x({0u, 2u}).build();
    CHECK(uu.interp_scalar(1u) == 7u);
    auto big = Interp1DBuilder<int>::new_(Array<int>::from_vec({-2000000000, 2000000000})).x({0, 1}).build();
    CHECK(throws<Panic>([&] { big.interp_scalar(1); }));
    auto ex = Interp1DBuilder<int>::new_(Array<int>::from_vec({0, 60000})).x({0, 1}).strategy(Linear<int>::new_().extrapolate(true)).build();
    CHECK(throws<Panic>([&] { ex.interp_scalar(60000); }));
  }
  // ---- integer element types: the reference's generic per-query path, no device ----
  {  // tests/interp2d.rs:29-47 (i32 data, default axes / i32 x axis), :63-82 (out of bounds, x before y)
    Array<int> d({3, 4}, std::vector<int>{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12});
    auto ip = Interp2DBuilder<int>::new_(d).build();
    CHECK(ip.interp_scalar(0, 0) == 1); CHECK(ip.interp_scalar(2, 3) == 12);
    CHECK(ip.interp_scalar(2, 0) == 9); CHECK(ip.interp_scalar(0, 3) == 4);
    auto ipx = Interp2DBuilder<int>::new_(d).x({1, 2, 3}).build();
    CHECK(ipx.interp_scalar(1, 0) == 1); CHECK(ipx.interp_scalar(3, 3) == 12);
    CHECK(ipx.interp_scalar(3, 0) == 9); CHECK(ipx.interp_scalar(1, 3) == 4);
    try { ip.interp(-1, 1); CHECK(false); } catch (const InterpolateError& e) { CHECK(e.axis == 0 && std::string(e.what()) == "x = -1 is not in range"); }
    try { ip.interp(1, -1); CHECK(false); } catch (const InterpolateError& e) { CHECK(e.axis == 1); }
    try { ip.interp(3, 1); CHECK(false); } catch (const InterpolateError& e) { CHECK(e.axis == 0); }
    try { ip.interp(1, 4); CHECK(false); } catch (const InterpolateError& e) { CHECK(e.axis == 1); }
    // batched entry = the trait's default loop; first-error index
    Array<int> qx({3}, std::vector<int>{0, 5, 1}), qy({3}, std::vector<int>{0, 0, 1});
    try { ip.interp_array(qx, qy); CHECK(false); } catch (const InterpolateError& e) { CHECK(e.index == 1); }
  }
  {  // Linear::calc_frac on i32 (linear.rs:29-36): integer division truncates toward zero
    auto ip = Interp1DBuilder<int>::new_(Array<int>::from_vec({0, 10})).x({0, 4}).build();
    auto r = ip.interp_array(Array<int>::from_vec({0, 1, 2, 3, 4}));
    CHECK((r.data == std::vector<int>{0, 2, 4, 6, 8}));
    auto ipn = Interp1DBuilder<int>::new_(Array<int>::from_vec({10, 0})).x({0, 4}).build();
    CHECK(ipn.interp_scalar(1) == 8 && ipn.interp_scalar(3) == 4);
    Array<int> d2({3, 2}, std::vector<int>{0, 7, 3, -7, 9, 0});
    auto ip2 = Interp1DBuilder<int>::new_(d2).x({0, 2, 5}).build();
    auto r2 = ip2.interp_array(Array<int>::from_vec({1, 2, 4}));
    CHECK((r2.data == std::vector<int>{1, 0, 3, -7, 7, -3}));
    auto ipe = Interp1DBuilder<long>::new_(Array<long>::from_vec({0, 10})).x({0, 4}).strategy(Linear<long>::new_().extrapolate(true)).build();
    CHECK(ipe.interp_scalar(6) == 12 && ipe.interp_scalar(-1) == -2);
    CHECK(throws<InterpolateError>([&] { ip.interp_scalar(5); }));
    // tests/interp1d.rs:122-140 with their i32 arrays
    CHECK(throws<BuilderError>([] { Interp1DBuilder<int>::new_(Array<int>::from_vec({1})).build(); }, BuilderError::NotEnoughData));
    CHECK(throws<BuilderError>([] { Interp1DBuilder<int>::new_(Array<int>::from_vec({1, 2})).x({1, 2, 3}).build(); }, BuilderError::ShapeError));
    CHECK(throws<BuilderError>([] { Interp1DBuilder<int>::new_(Array<int>::from_vec({1, 2, 3})).x({1, 2, 2}).build(); }, BuilderError::Monotonic));
  }
}

// examples/custom_strategy.rs
struct StepInterpolator : Interp1DStrategyBuilder<double>, Interp1DStrategy<double>,
                          std::enable_shared_from_this<StepInterpolator> {
  size_t MINIMUM_DATA_LENGHT() const override { return 2; }
  std::shared_ptr<Interp1DStrategy<double>> build(const std::vector<double>&, const Array<double>&) override {
    return std::make_shared<StepInterpolator>();
  }
  void interp_into(const Interp1D<double>& ip, double* target, double x) const override {
    size_t idx = ip.get_index_left_of(x);
    auto [xl, dl] = ip.index_point(idx);
    auto [xr, dr] = ip.index_point(idx + 1);
    const double* src = ((xr - xl) / 2.0 > (x - xl)) ? dl : dr;
    for (size_t l = 0; l < ip.lanes(); ++l) target[l] = src[l];
  }
};

static void device() {
  const double EPS = std::numeric_limits<double>::epsilon();
  {  // tests/interp1d.rs:21-30 interp_y_only
    auto ip = Interp1DBuilder<double>::new_(arr({1.5, 2.0, 3.0, 4.0, 5.0, 7.0, 7.0, 8.0, 9.0, 10.5})).build();
    CHECK(ip.interp_scalar(0.0) == 1.5); CHECK(ip.interp_scalar(9.0) == 10.5); CHECK(ip.interp_scalar(4.5) == 6.0);
    CHECK(ip.interp_scalar(0.25) == 1.625); CHECK(ip.interp_scalar(8.75) == 10.125);
  }
  {  // tests/interp1d.rs:72-80 extrapolate_with_x_and_y
    auto ip = Interp1DBuilder<double>::new_(arr({1.0, 0.0, 1.5})).x({0.0, 1.0, 1.5})
                  .strategy(Linear<double>::new_().extrapolate(true)).build();
    CHECK(ip.interp_scalar(-1.0) == 2.0); CHECK(ip.interp_scalar(2.0) == 3.0);
  }
  {  // tests/interp1d.rs:83-90 interp_array with a 2-D query
    auto ip = Interp1DBuilder<double>::new_(arr({1, 2, 3, 4, 5, 5, 4, 3, 2, 1})).build();
    Array<double> q({2, 3}, std::vector<double>{1.0, 2.0, 9.0, 4.0, 5.0, 7.5});
    auto r = ip.interp_array(q);
    CHECK((r.shape == std::vector<size_t>{2, 3}));
    CHECK((r.data == std::vector<double>{2.0, 3.0, 1.0, 5.0, 5.0, 2.5}));
  }
  {  // tests/interp1d.rs:93-120 out of bounds, message of linear.rs:81-83
    auto ip = Interp1DBuilder<double>::new_(arr({1, 2, 3})).build();
    CHECK(throws<InterpolateError>([&] { ip.interp(-0.1); }));
    CHECK(throws<InterpolateError>([&] { ip.interp(9.0); }));
    try { ip.interp_array(arr({0.5, 1.5, -0.1, 7.0})); CHECK(false); }
    catch (const InterpolateError& e) { CHECK(e.index == 2); CHECK(std::string(e.what()) == "x = -0.1 is not in range"); }
  }
  {  // tests/interp1d.rs:158-195 interp_multi_fn
    Array<double> data({4, 5}, std::vector<double>{0.1, 0.2, 0.3, 0.4, 0.5, 2, 2, 3, 4, 5, 10, 20, 30, 40, 50, 20, 40, 60, 80, 100});
    auto ip = Interp1DBuilder<double>::new_(data).x({1, 2, 3, 4}).build();
    auto r = ip.interp(1.5);
    const double e[5] = {1.05, 1.1, 1.65, 2.2, 2.75};
    for (int i = 0; i < 5; ++i) CHECK(std::fabs(r[i] - e[i]) <= EPS);
    Array<double> q({2, 2}, std::vector<double>{1.0, 1.5, 3.5, 4.0});
    auto rr = ip.interp_array(q);
    CHECK((rr.shape == std::vector<size_t>{2, 2, 5}));
    CHECK(std::fabs(rr[15] - 20.0) <= EPS && std::fabs(rr[19] - 100.0) <= EPS && std::fabs(rr[10] - 15.0) <= EPS);
    Array<double> bad({4});
    CHECK(throws<Panic>([&] { ip.interp_into(1.5, bad); }));
  }
  {  // cubic_spline.rs:62-82 doctest (abs f64::EPSILON)
    auto ip = Interp1DBuilder<double>::new_(arr({0.5, 0.0, 3.0})).strategy(CubicSpline<double>::new_()).x({-1.0, 0.0, 3.0}).build();
    auto r = ip.interp_array(Array<double>::linspace(-1.0, 3.0, 10));
    const double e[10] = {0.5, 0.1851851851851852, 0.01851851851851853, -5.551115123125783e-17, 0.12962962962962965,
                          0.40740740740740755, 0.8333333333333331, 1.407407407407407, 2.1296296296296293, 3.0};
    for (int i = 0; i < 10; ++i) CHECK(std::fabs(r[i] - e[i]) <= EPS);
  }
  {  // tests/cubic_spline_strat.rs:46-55, 442-452, 455-501 (first and last periodic values)
    auto ip = Interp1DBuilder<double>::new_(arr({1, 2, 1})).strategy(CubicSpline<double>::new_()).build();
    CHECK(throws<InterpolateError>([&] { ip.interp(-0.5); }));
    CHECK(throws<InterpolateError>([&] { ip.interp(3.5); }));
    Array<double> y({3, 2}, std::vector<double>{0.5, 1.0, 0.0, 1.5, 0.5, 1.1});
    CHECK(throws<BuilderError>([&] { Interp1DBuilder<double>::new_(y).strategy(
        CubicSpline<double>::new_().boundary(BoundaryCondition::Periodic())).build(); }, BuilderError::ValueError));
    auto per = Interp1DBuilder<double>::new_(arr({1.0, 2.0, 2.5, 2.5, 3.0, 2.0, 1.0, -2.0, 3.0, 5.0, 6.3, 1.0}))
                   .strategy(CubicSpline<double>::new_().extrapolate(true).boundary(BoundaryCondition::Periodic())).build();
    auto r = per.interp_array(Array<double>::linspace(-3.0, 15.0, 30));
    CHECK(std::fabs(r[0] - 3.0) <= 1e-3 * 3.0 && std::fabs(r[29] - 3.0) <= 1e-3 * 3.0 && std::fabs(r[1] - 4.45171164) <= 1e-3 * 4.45);
    auto d1 = Interp1DBuilder<double>::new_(arr({1.0, 2.0, 2.5, 2.5, 3.0, 2.0, 1.0, -2.0, 3.0, 5.0, 6.3, 8.0}))
                  .strategy(CubicSpline<double>::new_().extrapolate(true).boundary(
                      BoundaryCondition::Mixed(SingleBoundary::FirstDeriv(-0.1), SingleBoundary::FirstDeriv(-0.5)))).build();
    auto r1 = d1.interp_array(Array<double>::linspace(-3.0, 15.0, 30));
    CHECK(std::fabs(r1[0] - 45.12263976) <= 1e-3 * 45.2 && std::fabs(r1[29] + 165.7395108) <= 1e-3 * 166.0);
  }
  {  // examples/custom_strategy.rs:56-68 -- user strategy through the default batched hook
    auto ip = Interp1DBuilder<double>::new_(arr({2.0, 4.0, 5.0})).strategy(StepInterpolator()).build();
    auto r = ip.interp_array(Array<double>::linspace(-0.5, 2.5, 6));
    CHECK((r.data == std::vector<double>{2.0, 2.0, 4.0, 4.0, 5.0, 5.0}));
  }
  {  // tests/interp2d.rs:50-60, 63-82, 241-277
    Array<double> data({3, 4}, std::vector<double>{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12});
    auto ip = Interp2DBuilder<double>::new_(data).y({-3.0, -2.0, -1.0, 0.0}).build();
    CHECK(ip.interp_scalar(0.0, -3.0) == 1.0); CHECK(ip.interp_scalar(2.0, 0.0) == 12.0);
    CHECK(ip.interp_scalar(2.0, -3.0) == 9.0); CHECK(ip.interp_scalar(0.0, 0.0) == 4.0);
    auto ip2 = Interp2DBuilder<double>::new_(data).build();
    try { ip2.interp(1.0, -1.0); CHECK(false); } catch (const InterpolateError& e) { CHECK(e.axis == 1); }
    try { ip2.interp(3.0, 9.0); CHECK(false); } catch (const InterpolateError& e) { CHECK(e.axis == 0); }
    CHECK(throws<Panic>([&] { ip2.interp_array(arr({0.0, 1.0}), arr({0.0, 1.0, 2.0})); }));
    {  // inherent helpers of Interp2D (interp2d/mod.rs:150-167, 215-285, 348-372)
      auto pt = ip.index_point(1, 2);
      CHECK(pt.x == 1.0 && pt.y == -1.0 && pt.data[0] == 7.0);
      CHECK((ip.get_index_left_of(1.5, -2.5) == std::pair<size_t, size_t>{1, 0}));
      Array<double> one(std::vector<size_t>{}), wrong({2});
      ip.interp_into(2.0, 0.0, one);
      CHECK(one[0] == 12.0);
      CHECK(throws<Panic>([&] { ip.interp_into(2.0, 0.0, wrong); }));
      Array<double> buf({2}), bad({3});
      ip2.interp_array_into(arr({0.0, 2.0}), arr({0.0, 3.0}), buf);
      CHECK(buf[0] == 1.0 && buf[1] == 12.0);
      CHECK(throws<Panic>([&] { ip2.interp_array_into(arr({0.0, 2.0}), arr({0.0, 3.0}), bad); }));
    }
    Array<double> nd({2, 2, 2, 2}, std::vector<double>{1, 10, -1, -10, 2, 20, -2, -20, 3, 30, -3, -30, 5, 50, -5, -50});
    auto ip3 = Interp2DBuilder<double>::new_(nd).build();
    auto r = ip3.interp_array(arr({0.0, 0.5}), arr({0.5, 1.0}));
    CHECK((r.shape == std::vector<size_t>{2, 2, 2}));
    CHECK((r.data == std::vector<double>{1.5, 15.0, -1.5, -15.0, 3.5, 35.0, -3.5, -35.0}));
  }
  {  // interp_array through the library-owned device ring: every chunk equals the matching rows of one batch
    const size_t n = 50, L = 512, Q = 5000, chunk = 1024;
    std::vector<double> xk(n), yv(n * L), qs(Q);
    for (size_t i = 0; i < n; ++i) xk[i] = 0.1 * i + 0.01 * (i % 3);
    for (size_t i = 0; i < n * L; ++i) yv[i] = std::sin(0.001 * i) + 0.5;
    for (size_t i = 0; i < Q; ++i) qs[i] = xk[0] + (xk[n - 1] - xk[0]) * ((i * 7919u) % Q) / double(Q);
    auto ip = Interp1DBuilder<double>::new_(Array<double>({n, L}, yv)).x(xk).strategy(CubicSpline<double>::new_()).build();
    auto whole = ip.interp_array(Array<double>::from_vec(qs));
    std::vector<double> got(Q * L, -1.0);
    size_t chunks = 0;
    ip.interp_array_ring(Array<double>::from_vec(qs), chunk, 3, [&](const RingChunk& c) -> void* {
      CHECK(c.index == chunks && c.q_begin == chunks * chunk && c.row_stride == 3 * L && c.slot == chunks % 3);
      CHECK(hipMemcpy2DAsync(got.data() + c.q_begin * L, L * 8, c.out, c.row_stride * 8, L * 8, c.q_count, 2, c.stream) == 0);
      CHECK(hipStreamSynchronize(c.stream) == 0);
      ++chunks;
      return nullptr;
    });
    CHECK(chunks == 5);
    CHECK(got == whole.data);
    // first error: exactly the rows before it are produced
    qs[2500] = 99.0;
    size_t rows = 0;
    try {
      ip.interp_array_ring(Array<double>::from_vec(qs), chunk, 2, [&](const RingChunk& c) -> void* { rows += c.q_count; return nullptr; });
      CHECK(false);
    } catch (const InterpolateError& e) { CHECK(e.index == 2500 && rows == 2500); }
    // resident locator = one-shot search
    Locator<double> loc(xk);
    std::vector<double> lq{-1.0, xk[0], xk[7], 0.5 * (xk[7] + xk[8]), xk[n - 1], 100.0};
    CHECK(loc.get_lower_index(lq) == get_lower_index<double>(xk, lq));
    CHECK((loc.get_lower_index(lq) == std::vector<int64_t>{0, 0, 7, 7, (int64_t)n - 2, (int64_t)n - 2}));
  }
  {  // tests/cubic_spline_strat.rs:191-255 multidim_multi_bounds: BoundaryCondition::Individual, non-uniform x
    Array<double> y({3, 2}, std::vector<double>{0.5, 1.0, 0.0, 1.5, 3.0, 0.5});
    auto bc = BoundaryCondition::Individual({1, 2}, {RowBoundary::Natural(),
                                                     RowBoundary::Mixed(SingleBoundary::NotAKnot(), SingleBoundary::FirstDeriv(0.5))});
    auto ip = Interp1DBuilder<double>::new_(y).x({-1.0, 0.0, 3.0})
                  .strategy(CubicSpline<double>::new_().boundary(bc).extrapolate(true)).build();
    auto r = ip.interp_array(Array<double>::linspace(-2.0, 4.0, 15));
    const double e0[15] = {1., 0.85787172, 0.59766764, 0.30794461, 0.07725948, -0.00655977, 0.10058309, 0.375,
                           0.78717201, 1.30758017, 1.90670554, 2.55502915, 3.22303207, 3.88119534, 4.5};
    const double e1[15] = {-1.13194444, 0.02834467, 0.81235828, 1.27749433, 1.48115079, 1.48072562, 1.33361678,
                           1.09722222, 0.82893991, 0.5861678, 0.42630385, 0.40674603, 0.58489229, 1.01814059, 1.76388889};
    for (int i = 0; i < 15; ++i) {
      CHECK(std::fabs(r[2 * i] - e0[i]) <= 1e-3 * std::max(std::fabs(e0[i]), std::fabs(r[2 * i])) + EPS);
      CHECK(std::fabs(r[2 * i + 1] - e1[i]) <= 1e-3 * std::max(std::fabs(e1[i]), std::fabs(r[2 * i + 1])) + EPS);
    }
    // wrong shape of the boundary array: tests/cubic_spline_strat.rs:413-440
    CHECK(throws<BuilderError>([&] {
      Interp1DBuilder<double>::new_(y).strategy(CubicSpline<double>::new_().boundary(BoundaryCondition::Individual(
          {1, 3}, {RowBoundary::Natural(), RowBoundary::Clamped(), RowBoundary::NotAKnot()}))).build(); },
      BuilderError::ShapeError));
  }
  {  // several replicas, one call: contiguous blocks of the batch per replica (benches/bench_interp1d.rs:49-79 is the
     // reference's multi-worker shape); one replica per device when the box has several, else all on device 0
    const size_t n = 60, L = 256, Q = 9001;
    std::vector<double> xk(n), yv(n * L), qs(Q);
    for (size_t i = 0; i < n; ++i) xk[i] = 0.1 * i + 0.013 * (i % 4);
    for (size_t i = 0; i < n * L; ++i) yv[i] = std::cos(0.002 * i) + 0.25;
    for (size_t i = 0; i < Q; ++i) qs[i] = xk[0] + (xk[n - 1] - xk[0]) * ((i * 7907u) % Q) / double(Q);
    const int ndev = device_count();
    CHECK(current_device() == 0);
    CHECK(throws<DeviceError>([&] { set_current_device(ndev); }));
    std::vector<Interp1D<double>> reps;
    for (int r = 0; r < 3; ++r)
      reps.push_back(Interp1DBuilder<double>::new_(Array<double>({n, L}, yv)).x(xk)
                         .strategy(CubicSpline<double>::new_().device(ndev >= 3 ? r : 0)).build());
    std::vector<const Interp1D<double>*> rp{&reps[0], &reps[1], &reps[2]};
    CHECK(std::dynamic_pointer_cast<detail::Device1D<double>>(reps[2].strategy)->device == (ndev >= 3 ? 2 : 0));
    auto whole = reps[0].interp_array(Array<double>::from_vec(qs));
    auto sharded = interp_array_sharded(rp, Array<double>::from_vec(qs));
    CHECK(sharded.shape == whole.shape && sharded.data == whole.data);
    CHECK((shard_bounds(Q, 0, 3) == std::pair<size_t, size_t>{0, 3001}) && (shard_bounds(Q, 2, 3) == std::pair<size_t, size_t>{6001, 9001}));
    // first error over the whole batch: failures in shards 2 and 1 -> the lower one, rows after it untouched
    qs[7000] = -4.0; qs[4000] = 88.0;
    Array<double> buf({Q, L}, -9.0);
    try { interp_array_into_sharded(rp, Array<double>::from_vec(qs), buf); CHECK(false); }
    catch (const InterpolateError& e) { CHECK(e.index == 4000 && e.value == 88.0 && std::string(e.what()) == "x = 88.0 is not in range"); }
    CHECK(std::equal(buf.data.begin(), buf.data.begin() + 4000 * L, whole.data.begin()));
    CHECK(std::all_of(buf.data.begin() + 4000 * L, buf.data.end(), [](double v) { return v == -9.0; }));
    // ring, sharded: every chunk lands at its global position
    qs[7000] = xk[3]; qs[4000] = xk[5];
    auto whole2 = reps[1].interp_array(Array<double>::from_vec(qs));
    std::vector<double> got(Q * L, -1.0);
    std::mutex mu;
    size_t rows = 0;
    interp_array_ring_sharded<double>(rp, Array<double>::from_vec(qs), 1024, 2, [&](const RingChunk& c) -> void* {
      CHECK(c.shard < 3 && c.q_begin >= shard_bounds(Q, c.shard, 3).first && c.q_begin + c.q_count <= shard_bounds(Q, c.shard, 3).second);
      CHECK(hipMemcpy2DAsync(got.data() + c.q_begin * L, L * 8, c.out, c.row_stride * 8, L * 8, c.q_count, 2, c.stream) == 0);
      CHECK(hipStreamSynchronize(c.stream) == 0);
      std::lock_guard<std::mutex> g(mu);
      rows += c.q_count;
      return nullptr;
    });
    CHECK(rows == Q && got == whole2.data);
    // replicas by device-to-device copy of the tables (ndi_interp1d_clone) behave like rebuilt ones
    {
      std::vector<Interp1D<double>> cl;
      for (int r = 0; r < 2; ++r) cl.push_back(reps[0].replicate(ndev >= 2 ? r : 0));
      std::vector<const Interp1D<double>*> cp{&cl[0], &cl[1]};
      CHECK(interp_array_sharded(cp, Array<double>::from_vec(qs)).data == whole2.data);
      CHECK(std::dynamic_pointer_cast<detail::Device1D<double>>(cl[1].strategy)->device == (ndev >= 2 ? 1 : 0));
    }
    // a worker thread that owns a device: set_current_device once, then unchanged builder code
    set_current_device(ndev - 1);
    auto lin = Interp1DBuilder<double>::new_(arr({1, 2, 3})).build();
    CHECK(std::dynamic_pointer_cast<detail::Device1D<double>>(lin.strategy)->device == ndev - 1);
    CHECK(lin.interp_scalar(1.5) == 2.5);
    set_current_device(0);
    // 2-D
    Array<double> g2({5, 4, 3});
    for (size_t i = 0; i < g2.len(); ++i) g2[i] = 0.25 * double((i * 37) % 19);
    std::vector<Interp2D<double>> r2;
    for (int r = 0; r < 2; ++r)
      r2.push_back(Interp2DBuilder<double>::new_(g2).strategy(Bilinear<double>::new_().device(ndev >= 2 ? r : 0)).build());
    std::vector<const Interp2D<double>*> rp2{&r2[0], &r2[1]};
    Array<double> qx({7}), qy({7});
    for (size_t i = 0; i < 7; ++i) { qx[i] = 4.0 * i / 6.0; qy[i] = 3.0 * ((i * 5) % 7) / 6.0; }
    auto w2 = r2[0].interp_array(qx, qy);
    CHECK(interp_array_sharded(rp2, qx, qy).data == w2.data);
    CHECK(r2[1].replicate(0).interp_array(qx, qy).data == w2.data);
    qy[5] = 3.5;
    try { interp_array_sharded(rp2, qx, qy); CHECK(false); } catch (const InterpolateError& e) { CHECK(e.index == 5 && e.axis == 1); }
  }
  {  // f32 is a first-class type (tests/cubic_spline_strat.rs:108-154)
    Array<float> d = Array<float>::from_vec({1.f, 2.f, 2.5f, 2.5f, 3.f, 2.f, 1.f, -2.f, 3.f, 5.f, 6.3f, 8.f});
    auto ip = Interp1DBuilder<float>::new_(d).strategy(CubicSpline<float>::new_().extrapolate(true)).build();
    auto r = ip.interp_array(Array<float>::linspace(-3.f, 15.f, 30));
    CHECK(std::fabs(r[0] - 0.94398816f) <= 1e-3f && std::fabs(r[29] - 5.825231f) <= 6e-3f);
  }
}

int main(int argc, char** argv) {
  const bool host_only_flag = argc > 1 && std::strcmp(argv[1], "--host-only") == 0;
  host_only();
  if (!host_only_flag) device();
  std::printf("%s: %d failed check(s)\n", host_only_flag ? "host-only" : "host+device", failures);
  return failures;
}
