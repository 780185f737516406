"""Generic per-query strategies for element types the device path does not cover (integers, f16 ...).

The reference's `Linear` and `Bilinear` are generic over `T: Num + PartialOrd + ...`
(src/interp1d/strategies/linear.rs:13-20, src/interp2d/strategies/bilinear.rs:20-27), so its own tests
interpolate `i32` data on `i32` axes (tests/interp1d.rs:122-140, tests/interp2d.rs:14-61).  The MI355X kernels
cover f32 / f64 only; every other element type takes the strategy trait's *default* batched
hook -- the reference's serial query loop (interp1d/mod.rs:326-343) -- over the per-query bodies below, which
restate `interp_into` with the element type's own arithmetic (integer division truncates toward zero, as
`i32 / i32` does in Rust).  f32 / f64 never come here: for them the HIP library is the only path.
"""
from __future__ import annotations

import math

import numpy as np

from .errors import InterpolateError, Panic


def _is_int(dt) -> bool:
    return np.dtype(dt).kind in "iu"


def _div(a, b, integer: bool):
    if not integer:
        return a / b
    if b == 0:
        raise Panic("attempt to divide by zero")
    q = abs(a) // abs(b)            # Rust integer division truncates toward zero
    return q if (a < 0) == (b < 0) else -q


class _IntRange:
    """Every intermediate of the integer path must be representable in the element type T, as in the reference's
    `T` arithmetic: a Rust debug build panics with "attempt to subtract / multiply / add with overflow" (a release
    build wraps; results that only exist through wrapping are not reproduced -- the panic is)."""

    def __init__(self, dt):
        info = np.iinfo(dt)
        self.lo, self.hi = int(info.min), int(info.max)

    def check(self, v: int, op: str) -> int:
        if v < self.lo or v > self.hi:
            raise Panic(f"attempt to {op} with overflow")
        return v


def calc_frac(p1, p2, x, integer: bool, rng: "_IntRange | None" = None):
    """Linear::calc_frac (linear.rs:29-36): m = (y2 - y1) / (x2 - x1); m * (x - x1) + y1."""
    (x1, y1), (x2, y2) = p1, p2
    if not integer or rng is None:
        m = _div(y2 - y1, x2 - x1, integer)
        return m * (x - x1) + y1
    dy = rng.check(y2 - y1, "subtract")
    dx = rng.check(x2 - x1, "subtract")
    m = rng.check(_div(dy, dx, True), "divide")          # i32::MIN / -1
    d = rng.check(x - x1, "subtract")
    return rng.check(rng.check(m * d, "multiply") + y1, "add")


def _scalar(v, dt):
    """Element-type arithmetic: Python ints for integer types (exact, range-checked by the caller), numpy scalars
    otherwise.  A query that is not a value of the integer element type (2.7 for i32 data) is a type error in
    the reference (the query type IS the element type), so it is refused here instead of being truncated."""
    if not _is_int(dt):
        return np.dtype(dt).type(v)
    if isinstance(v, (float, np.floating)) and (not math.isfinite(float(v)) or float(v) != int(v)):
        raise TypeError(f"query {v!r} is not a value of the element type {np.dtype(dt)}")
    iv = int(v)
    info = np.iinfo(dt)
    if iv < info.min or iv > info.max:
        raise TypeError(f"query {v!r} is out of range for the element type {np.dtype(dt)}")
    return iv


def lower_index(knots, x, integer: bool) -> int:
    """VectorExtensions::get_lower_index (src/vector_extensions.rs:55-111): the unique i with
    k[i] <= x < k[i+1], clamped to [0, n-2]; NaN panics (:83-84)."""
    n = len(knots)
    if not integer and x != x:
        raise Panic("not implemented: failed to convert NaN to usize")
    if x <= knots[0]:
        return 0
    if x >= knots[n - 1]:
        return n - 2
    lo, hi = 0, n - 1               # invariant k[lo] <= x < k[hi]  (:100-110)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if knots[mid] <= x:
            lo = mid
        else:
            hi = mid
    return lo


def _debug(v, integer: bool) -> str:
    from .errors import _rust_float
    return str(int(v)) if integer else _rust_float(float(v))


class HostLinear:
    """Linear::interp_into (linear.rs:73-98) for a non-f32/f64 element type."""

    def __init__(self, x, data, extrapolate: bool):
        self._dt = np.asarray(data).dtype
        self._int = _is_int(self._dt)
        self._rng = _IntRange(self._dt) if self._int else None
        self._x = [_scalar(v, self._dt) for v in np.asarray(x).reshape(-1)]
        self._rows = np.asarray(data).reshape(len(self._x), -1)
        self._extrapolate = bool(extrapolate)

    def interp_into(self, interpolator, target, x):
        x = _scalar(x, self._dt)
        k = self._x
        if not self._extrapolate and not (k[0] <= x <= k[-1]):
            raise InterpolateError.OutOfBounds(f"x = {_debug(x, self._int)} is not in range", value=x, axis=0)
        i = lower_index(k, x, self._int)
        y1, y2 = self._rows[i], self._rows[i + 1]
        res = np.empty(y1.size, dtype=self._dt)
        for l in range(res.size):
            res[l] = calc_frac((k[i], _scalar(y1[l], self._dt)), (k[i + 1], _scalar(y2[l], self._dt)), x, self._int,
                               self._rng)
        target[...] = res.reshape(target.shape)      # target may be a strided view

    # trait default: the reference's serial query loop, stopping at the first Err (interp1d/mod.rs:326-343)
    def interp_array_into(self, interpolator, xs_flat, out2d, **_kw):
        for i in range(len(xs_flat)):
            try:
                self.interp_into(interpolator, out2d[i], xs_flat[i])
            except InterpolateError.OutOfBounds as e:
                e.index = i
                raise

    def release(self):
        pass


class HostBilinear:
    """Bilinear::interp_into (bilinear.rs:64-99) for a non-f32/f64 element type."""

    def __init__(self, x, y, data, extrapolate: bool):
        d = np.asarray(data)
        self._dt = d.dtype
        self._int = _is_int(self._dt)
        self._rng = _IntRange(self._dt) if self._int else None
        self._x = [_scalar(v, self._dt) for v in np.asarray(x).reshape(-1)]
        self._y = [_scalar(v, self._dt) for v in np.asarray(y).reshape(-1)]
        self._grid = d.reshape(len(self._x), len(self._y), -1)
        self._extrapolate = bool(extrapolate)

    def interp_into(self, interpolator, target, x, y):
        x, y = _scalar(x, self._dt), _scalar(y, self._dt)
        kx, ky = self._x, self._y
        if not self._extrapolate and not (kx[0] <= x <= kx[-1]):       # x before y, bilinear.rs:71-80
            raise InterpolateError.OutOfBounds(f"x = {_debug(x, self._int)} is not in range", value=x, axis=0)
        if not self._extrapolate and not (ky[0] <= y <= ky[-1]):
            raise InterpolateError.OutOfBounds(f"y = {_debug(y, self._int)} is not in range", value=y, axis=1)
        xi, yi = lower_index(kx, x, self._int), lower_index(ky, y, self._int)
        x1, x2, y1, y2 = kx[xi], kx[xi + 1], ky[yi], ky[yi + 1]
        g, s = self._grid, (lambda v: _scalar(v, self._dt))
        res = np.empty(g.shape[2], dtype=self._dt)
        for l in range(res.size):
            z11, z12, z21, z22 = s(g[xi, yi, l]), s(g[xi, yi + 1, l]), s(g[xi + 1, yi, l]), s(g[xi + 1, yi + 1, l])
            z1 = calc_frac((x1, z11), (x2, z21), x, self._int, self._rng)           # bilinear.rs:88-97
            z2 = calc_frac((x1, z12), (x2, z22), x, self._int, self._rng)
            res[l] = calc_frac((y1, z1), (y2, z2), y, self._int, self._rng)
        target[...] = res.reshape(target.shape)

    def interp_array_into(self, interpolator, xs_flat, ys_flat, out2d, **_kw):
        for i in range(len(xs_flat)):
            try:
                self.interp_into(interpolator, out2d[i], xs_flat[i], ys_flat[i])
            except InterpolateError.OutOfBounds as e:
                e.index = i
                raise

    def release(self):
        pass
