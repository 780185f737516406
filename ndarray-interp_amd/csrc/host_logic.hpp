// csrc/host_logic.hpp -- the lane-independent, O(n) host side of the build:
//   * axis validation (Interp1DBuilder::build / Interp2DBuilder::build),
//   * the monotonicity scan (VectorExtensions::monotonic_prop),
//   * the spline's tridiagonal plan: the three diagonals and the Thomas elimination
//     factors depend on x only (cubic_spline.rs:431-451, 691-692), so they are formed
//     once here in T precision, in the reference's operation order, and shared by every
//     lane of the device solve.
// Compiled with -ffp-contract=off like the device code: rustc never fuses a*b+c.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

#include "common.hpp"

namespace ndi {

// VectorExtensions::monotonic_prop, src/vector_extensions.rs:40-53 with the
// MonotonicState machine of :116-198 folded into two flags.
template <class T>
int monotonic_scan(const T* v, uint64_t n) {
  if (n <= 1) return NDI_MONO_NOT;
  int dir = 0;         // 0 undecided (only equal pairs so far), +1 rising, -1 falling
  bool strict = true;  // no equal pair seen yet
  for (uint64_t i = 0; i + 1 < n; ++i) {
    const T a = v[i], b = v[i + 1];
    if (a == b) {
      strict = false;
      continue;
    }
    // a NaN compares false everywhere: `a < b` false, `a == b` false -> treated as a fall
    // while rising (-> NotMonotonic) exactly like the reference's else-branches.
    const int step = (a < b) ? +1 : -1;
    if (dir == 0) {
      dir = step;
    } else if (dir == +1) {
      if (!(a < b)) return NDI_MONO_NOT;
    } else {
      if (!(a > b)) return NDI_MONO_NOT;
    }
  }
  if (dir == 0) return NDI_MONO_NOT;  // all equal: NotStrict -> NotMonotonic (:194)
  if (dir > 0) return strict ? NDI_MONO_RISING_STRICT : NDI_MONO_RISING;
  return strict ? NDI_MONO_FALLING_STRICT : NDI_MONO_FALLING;
}

inline uint64_t min_len_1d(int strategy) {
  // MINIMUM_DATA_LENGHT: Linear 2 (linear.rs:52), CubicSpline 3 (cubic_spline.rs:751)
  return strategy == NDI_CUBIC_SPLINE ? 3 : 2;
}

// Interp1DBuilder::build, src/interp1d/mod.rs:449-471 (check order preserved).
template <class T>
ndi_status check_axis_1d(const T* x, uint64_t x_len, uint64_t n, int strategy) {
  const uint64_t need = min_len_1d(strategy);
  if (n < need)
    return fail(NDI_NOT_ENOUGH_DATA,
                "The chosen Interpolation strategy needs at least %llu data points",
                (unsigned long long)need);
  if (monotonic_scan(x, x_len) != NDI_MONO_RISING_STRICT)
    return fail(NDI_MONOTONIC, "Values in the x axis need to be strictly monotonic rising");
  if (x_len != n)
    return fail(NDI_SHAPE, "Lengths of x and data axis need to match. Got x: %llu, data: %llu",
                (unsigned long long)x_len, (unsigned long long)n);
  return NDI_OK;
}

// Interp2DBuilder::build, src/interp2d/mod.rs:477-509 (check order preserved).
template <class T>
ndi_status check_axes_2d(const T* x, uint64_t x_len, const T* y, uint64_t y_len, uint64_t nx,
                         uint64_t ny) {
  const unsigned long long need = 2;  // Bilinear MINIMUM_DATA_LENGHT, bilinear.rs:41
  if (nx < need)
    return fail(NDI_NOT_ENOUGH_DATA,
                "The 0-dimension has not enough data for the chosen interpolation strategy. "
                "Provided: %llu, Reqired: %llu", (unsigned long long)nx, need);
  if (ny < need)
    return fail(NDI_NOT_ENOUGH_DATA,
                "The 1-dimension has not enough data for the chosen interpolation strategy. "
                "Provided: %llu, Reqired: %llu", (unsigned long long)ny, need);
  if (x_len != nx)
    return fail(NDI_SHAPE, "Lenghts of x-axis and data-0-axis need to match. Got x: %llu, data-0: %llu",
                (unsigned long long)x_len, (unsigned long long)nx);
  if (y_len != ny)
    return fail(NDI_SHAPE, "Lenghts of y-axis and data-1-axis need to match. Got y: %llu, data-1: %llu",
                (unsigned long long)y_len, (unsigned long long)ny);
  if (monotonic_scan(x, x_len) != NDI_MONO_RISING_STRICT)
    return fail(NDI_MONOTONIC, "The x-axis needs to be strictly monotonic rising");
  if (monotonic_scan(y, y_len) != NDI_MONO_RISING_STRICT)
    return fail(NDI_MONOTONIC, "The y-axis needs to be strictly monotonic rising");
  return NDI_OK;
}

// ---------------------------------------------------------------------------------------------
// Spline plan
// ---------------------------------------------------------------------------------------------
enum SplineMode : int {
  SPLINE_GENERAL = 0,     // n >= 3, Mixed{left,right} boundaries (cubic_spline.rs:597-670)
  SPLINE_PARABOLA3 = 1,   // n == 3 and NotAKnot on both ends (:569-596)
  SPLINE_PERIODIC = 2,    // n >= 4 periodic, condensed system (:498-565)
  SPLINE_PERIODIC3 = 3,   // n == 3 periodic, closed form (:480-496)
};

// Boundary kinds after SingleBoundary::specialize (:287-296): only these three remain.
enum EndKind : int { END_NOT_A_KNOT = 0, END_FIRST_DERIV = 1, END_SECOND_DERIV = 2 };

template <class T>
struct SplinePlan {
  int mode = SPLINE_GENERAL;
  uint64_t n = 0;
  uint64_t m = 0;             // order of the system handed to the Thomas sweeps
  std::vector<T> dx;          // dx[i] = x[i+1] - x[i], i < n-1
  std::vector<T> up;          // upper diagonal, m entries
  std::vector<T> w;           // elimination factors  w[i] = low[i] / mid'[i-1]
  std::vector<T> midp;        // eliminated main diagonal mid'[i]
  std::vector<T> k2;          // periodic only: solution of the lane-independent second system
  T per_den = T(0);           // periodic only: denominator of k_{n-2}
  int left_kind = END_NOT_A_KNOT, right_kind = END_NOT_A_KNOT;
  T left_val = T(0), right_val = T(0);
  // not-a-knot end rows (:599-611, :634-648)
  T nkL_tmp1 = T(0), nkL_d = T(1), nkR_tmp1 = T(0), nkR_d = T(1);
  T dx0_sq = T(0), dxl_sq = T(0);  // dx0.pow(2), dx_1.pow(2): plain products, see DESIGN.md
};

inline void specialize_end(int kind, double val, int& out_kind, double& out_val) {
  switch (kind) {
    case NDI_BC_NATURAL: out_kind = END_SECOND_DERIV; out_val = 0.0; break;
    case NDI_BC_CLAMPED: out_kind = END_FIRST_DERIV; out_val = 0.0; break;
    case NDI_BC_FIRST_DERIV: out_kind = END_FIRST_DERIV; out_val = val; break;
    case NDI_BC_SECOND_DERIV: out_kind = END_SECOND_DERIV; out_val = val; break;
    default: out_kind = END_NOT_A_KNOT; out_val = 0.0; break;
  }
}

// Forward elimination of CubicSpline::thomas on the diagonals alone (:690-692).
template <class T>
void eliminate(const std::vector<T>& up, std::vector<T>& mid, const std::vector<T>& low,
               std::vector<T>& w, std::vector<T>& midp) {
  const size_t m = mid.size();
  w.assign(m, T(0));
  midp = std::move(mid);   // (the caller is done with the uneliminated diagonal)
  for (size_t i = 1; i < m; ++i) {
    w[i] = low[i] / midp[i - 1];
    midp[i] -= w[i] * up[i - 1];
  }
}

// Full Thomas solve for one lane-independent right-hand side (used for the periodic k2).
template <class T>
std::vector<T> thomas_host(const std::vector<T>& up, const std::vector<T>& w,
                           const std::vector<T>& midp, std::vector<T> rhs) {
  const size_t m = rhs.size();
  for (size_t i = 1; i < m; ++i) rhs[i] = rhs[i] - w[i] * rhs[i - 1];
  std::vector<T> k(m);
  k[m - 1] = rhs[m - 1] / midp[m - 1];
  for (size_t i = m - 1; i-- > 0;) k[i] = (rhs[i] - up[i] * k[i + 1]) / midp[i];
  return k;
}

// The boundary-row scalars of the GENERAL plan alone (n >= 4, not periodic) -- what the device needs when it forms the
// diagonals and the elimination factors itself (spline_eliminate_kernel): no O(n) vector is built.  up_first / mid_first /
// low_last / mid_last are rows 0 and n-1 of the three diagonals (:597-670).
template <class T>
struct SplineEnds {
  T up_first = T(0), mid_first = T(0), low_last = T(0), mid_last = T(0);
};

template <class T>
SplinePlan<T> make_spline_plan_scalars(const T* x, uint64_t n, int lkind, double lval, int rkind, double rval,
                                       SplineEnds<T>& E) {
  SplinePlan<T> P;
  P.n = n;
  P.m = n;
  P.mode = SPLINE_GENERAL;
  const T one = T(1), two = T(2);
  const T dx0 = x[1] - x[0];
  const T dx1 = x[2] - x[1];
  const T dxl = x[n - 1] - x[n - 2];
  const T dxl2 = x[n - 2] - x[n - 3];
  P.dx0_sq = dx0 * dx0;
  P.dxl_sq = dxl * dxl;
  int lk, rk;
  double lv, rv;
  specialize_end(lkind, lval, lk, lv);
  specialize_end(rkind, rval, rk, rv);
  P.left_kind = lk;
  P.right_kind = rk;
  P.left_val = T(lv);
  P.right_val = T(rv);
  if (lk == END_NOT_A_KNOT) {
    E.mid_first = dx1;
    const T d = x[2] - x[0];
    E.up_first = d;
    P.nkL_d = d;
    P.nkL_tmp1 = (dx0 + two * d) * dx1;
  } else if (lk == END_FIRST_DERIV) {
    E.mid_first = one; E.up_first = T(0);
  } else {
    E.up_first = dx0; E.mid_first = two * dx0;
  }
  if (rk == END_NOT_A_KNOT) {
    E.mid_last = dxl;  // sic: the reference uses dx_1 here (:635)
    const T d = x[n - 1] - x[n - 3];
    E.low_last = d;
    P.nkR_d = d;
    P.nkR_tmp1 = (two * d + dxl) * dxl2;
  } else if (rk == END_FIRST_DERIV) {
    E.mid_last = one; E.low_last = T(0);
  } else {
    E.mid_last = two * dxl; E.low_last = dxl;
  }
  return P;
}

template <class T>
SplinePlan<T> make_spline_plan(const T* x, uint64_t n, bool periodic, int lkind, double lval,
                               int rkind, double rval) {
  SplinePlan<T> P;
  P.n = n;
  const T one = T(1), two = T(2);
  P.dx.resize(n - 1);
  for (uint64_t i = 0; i + 1 < n; ++i) P.dx[i] = x[i + 1] - x[i];
  const T dx0 = x[1] - x[0];
  const T dx1 = x[2] - x[1];
  const T dxl = x[n - 1] - x[n - 2];   // dx_1 in the reference
  const T dxl2 = x[n - 2] - x[n - 3];  // dx_2
  P.dx0_sq = dx0 * dx0;
  P.dxl_sq = dxl * dxl;

  // interior rows 1..n-2 (:440-451)
  std::vector<T> up(n, T(0)), mid(n, T(0)), low(n, T(0));
  for (uint64_t i = 1; i + 1 < n; ++i) {
    const T dxn = x[i + 1] - x[i];
    const T dxn_1 = x[i] - x[i - 1];
    up[i] = dxn_1;
    mid[i] = two * (dxn + dxn_1);
    low[i] = dxn;
  }

  if (periodic && n == 3) {
    P.mode = SPLINE_PERIODIC3;
    P.m = 0;
    return P;
  }
  if (periodic) {
    P.mode = SPLINE_PERIODIC;
    const uint64_t m = n - 2;
    up.resize(m);
    mid.resize(m);
    low.resize(m);
    mid[0] = two * (dxl + dx0);
    up[0] = dxl;
    P.m = m;
    P.up = std::move(up);
    eliminate(P.up, mid, low, P.w, P.midp);
    std::vector<T> rhs2(m, T(0));
    const T dx_3 = x[n - 3] - x[n - 4];
    rhs2[0] = -dx0;
    rhs2[m - 1] = -dx_3;
    P.k2 = thomas_host(P.up, P.w, P.midp, rhs2);
    P.per_den = P.k2[0] * dxl2 + P.k2[m - 1] * dxl + two * (dxl + dxl2);
    return P;
  }

  int lk, rk;
  double lv, rv;
  specialize_end(lkind, lval, lk, lv);
  specialize_end(rkind, rval, rk, rv);
  P.left_kind = lk;
  P.right_kind = rk;
  P.left_val = T(lv);
  P.right_val = T(rv);

  if (n == 3 && lk == END_NOT_A_KNOT && rk == END_NOT_A_KNOT) {
    P.mode = SPLINE_PARABOLA3;
    mid[0] = one; up[0] = one;
    low[1] = dx1; mid[1] = two * (dx0 + dx1); up[1] = dx0;
    low[2] = one; mid[2] = one;
  } else {
    P.mode = SPLINE_GENERAL;
    if (lk == END_NOT_A_KNOT) {
      mid[0] = dx1;
      const T d = x[2] - x[0];
      up[0] = d;
      P.nkL_d = d;
      P.nkL_tmp1 = (dx0 + two * d) * dx1;
    } else if (lk == END_FIRST_DERIV) {
      mid[0] = one; up[0] = T(0);
    } else {
      up[0] = dx0; mid[0] = two * dx0;
    }
    if (rk == END_NOT_A_KNOT) {
      mid[n - 1] = dxl;  // sic: the reference uses dx_1 here (:635)
      const T d = x[n - 1] - x[n - 3];
      low[n - 1] = d;
      P.nkR_d = d;
      P.nkR_tmp1 = (two * d + dxl) * dxl2;
    } else if (rk == END_FIRST_DERIV) {
      mid[n - 1] = one; low[n - 1] = T(0);
    } else {
      mid[n - 1] = two * dxl; low[n - 1] = dxl;
    }
  }
  P.m = n;
  P.up = std::move(up);
  eliminate(P.up, mid, low, P.w, P.midp);
  return P;
}

}  // namespace ndi
