"""Host-side mirror of the reference's 1-D surface (src/interp1d/mod.rs + strategies/):
`Interp1DBuilder`, `Interp1D`, the strategy trait pair and the built-in `Linear` and `CubicSpline`.

Names, argument meaning and error behaviour follow the reference so that the parity tests read like
its own tests.  The built-in strategies override the *batched* hook (`interp_array_into`) and call the
C ABI (include/ndinterp.h); the per-query hook `interp_into` -- the only thing the reference's trait has
(strategies/mod.rs:59-64) -- stays the extension point for user strategies, whose default batched hook
is the reference's own per-query loop (interp1d/mod.rs:326-343).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._arrays import OUTPUT_OWNED_MIN_BYTES, Buf, current_stream_ptr, dtype_id, is_torch, np_dtype_of, output_empty
from .errors import BuilderError, InterpolateError, Panic, raise_builder, raise_eval
from .vector_extensions import Monotonic, get_lower_index, monotonic_prop


# ------------------------------------------------------------------------------------------------
# strategy traits (src/interp1d/strategies/mod.rs:12-65)
# ------------------------------------------------------------------------------------------------
class Interp1DStrategyBuilder:
    """Trait `Interp1DStrategyBuilder`: `MINIMUM_DATA_LENGHT` (sic) and `build(x, data)`."""

    MINIMUM_DATA_LENGHT = 2

    def build(self, x, data) -> "Interp1DStrategy":
        raise NotImplementedError


class Interp1DStrategy:
    """Trait `Interp1DStrategy`: per-query `interp_into(interpolator, target, x)`.

    `interp_array_into` is the defaulted batched hook added by this build: the default body is the
    reference's serial query loop, so strategies that only implement `interp_into`
    (examples/custom_strategy.rs) keep working unchanged on the host."""

    def interp_into(self, interpolator: "Interp1D", target: np.ndarray, x) -> None:
        raise NotImplementedError

    def interp_array_into(self, interpolator: "Interp1D", xs_flat, out2d) -> None:
        for i in range(len(xs_flat)):  # Zip(xs, rows).fold_while, stops at the first Err
            self.interp_into(interpolator, out2d[i].reshape(interpolator.data.shape[1:]), xs_flat[i])

    def release(self) -> None:
        pass


class _DeviceStrategy1D(Interp1DStrategy):
    """Shared body of the built-in strategies: owns an `ndi_interp1d*`."""

    _kind = _capi.LINEAR
    path = _capi.PATH_AUTO  # evaluation formulation, see include/ndinterp.h ndi_path

    def __init__(self):
        self._h = None
        self._device = 0
        self._np_dtype = None
        self._lanes = 1
        self._inflight = []

    # -- build ------------------------------------------------------------------------------
    def _create(self, x, data, *, extrapolate, periodic=False, left=(0, 0.0), right=(0, 0.0),
                per_lane=None, device=None, build_flags=0):
        xb_dt = np_dtype_of(data)
        tid = dtype_id(xb_dt)
        db = Buf(data)
        xb = None
        if x is not None:   # the axis travels in the same memory space as the data
            xb = Buf(_host(x), xb_dt) if db.memspace == _capi.MEM_HOST else Buf(_to_device(x, db.keep.device), xb_dt)
        if device is None:
            device = db.device if db.memspace == _capi.MEM_DEVICE else _default_device()
        n = db.shape[0]
        lanes = int(np.prod(db.shape[1:], dtype=np.int64)) if len(db.shape) > 1 else 1
        d = _capi.Interp1DDesc()
        d.dtype, d.strategy, d.extrapolate, d.device = tid, self._kind, int(bool(extrapolate)), device
        d.n, d.lanes = n, lanes
        d.x_len = xb.size if xb is not None else n
        d.x = xb.ptr if xb is not None else None
        d.data = db.ptr
        d.memspace = db.memspace
        d.validate = 0  # Interp1DBuilder.build() has validated already, as in the reference (:449-473)
        d.periodic = int(bool(periodic))
        d.build_flags = int(build_flags)
        d.left = _capi.Boundary(int(left[0]), float(left[1]))
        d.right = _capi.Boundary(int(right[0]), float(right[1]))
        keep = []
        if per_lane is not None:
            lk, lv, rk, rv = per_lane
            lk = np.ascontiguousarray(lk, np.int32); rk = np.ascontiguousarray(rk, np.int32)
            lv = np.ascontiguousarray(lv, np.float64); rv = np.ascontiguousarray(rv, np.float64)
            keep = [lk, lv, rk, rv]
            d.lane_left_kind, d.lane_left_value = lk.ctypes.data, lv.ctypes.data
            d.lane_right_kind, d.lane_right_value = rk.ctypes.data, rv.ctypes.data
        h = C.c_void_p()
        st = _capi.lib().ndi_interp1d_create(C.byref(d), C.byref(h))
        del keep
        if st != _capi.OK:
            raise_builder(st)
        self._h, self._device, self._np_dtype, self._lanes = h, device, xb_dt, lanes
        return self

    def release(self):
        if self._h is not None:
            _capi.lib().ndi_interp1d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def clone(self, device: int):
        """A replica of this finished strategy on `device` (ndi_interp1d_clone): tables copied device to device."""
        import copy
        h = C.c_void_p()
        st = _capi.lib().ndi_interp1d_clone(self._h, int(device), C.byref(h))
        if st != _capi.OK:
            raise_builder(st)
        other = copy.copy(self)
        other._h, other._device, other._inflight = h, int(device), []
        return other

    # -- evaluate -----------------------------------------------------------------------------
    _takes_fresh = True   # interp_array() may tell this strategy that the output buffer is its own (ndi_eval_flags)

    def interp_array_into(self, interpolator, xs_flat, out2d, *, async_launch=False, fresh=False,
                          rows_after_error_unspecified=False):
        """Replaces the reference's query loop (interp1d/mod.rs:326-343) by one C-ABI call.  `fresh`: the buffer was
        allocated for this call and is dropped on Err (Interp1D::interp_array, :197-211) -- NDI_EVAL_FRESH_OUTPUT."""
        qb = Buf(xs_flat, self._np_dtype)
        _check_out_dtype(out2d, self._np_dtype)
        opts = _capi.EvalOpts()
        opts.q_memspace = qb.memspace
        opts.path = self.path
        opts.async_launch = int(bool(async_launch))
        # rows_after_error_unspecified: a caller-owned buffer whose rows at / after a failing query the caller gives up
        # (NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED; the reference leaves them untouched, interp1d/mod.rs:334-342)
        opts.flags = (_capi.EVAL_FRESH_OUTPUT if fresh else _capi.EVAL_DEFAULT) | \
            (_capi.EVAL_ROWS_AFTER_ERROR_UNSPECIFIED if rows_after_error_unspecified else 0)
        # an async batch reads the query array until finish(): keep every (possibly converted) copy alive
        if async_launch:
            self._inflight.append(qb)
        if is_torch(out2d):
            if not out2d.is_cuda:
                raise TypeError("torch output buffers must live on the device; use numpy for host buffers")
            opts.out_memspace = _capi.MEM_DEVICE
            optr = out2d.data_ptr()
            stride = out2d.stride(0) if out2d.dim() > 1 and out2d.shape[0] > 1 else self._lanes
            opts.stream = current_stream_ptr(self._device)
        else:
            opts.out_memspace = _capi.MEM_HOST
            optr = out2d.ctypes.data
            stride = out2d.strides[0] // out2d.itemsize if out2d.ndim > 1 and out2d.shape[0] > 1 else self._lanes
            if qb.memspace == _capi.MEM_DEVICE:
                opts.stream = current_stream_ptr(self._device)
        info = _capi.OobInfo()
        st = _capi.lib().ndi_interp1d_eval(self._h, qb.ptr, qb.size, optr, max(stride, self._lanes),
                                           C.byref(opts), C.byref(info))
        if st != _capi.OK:
            raise_eval(st, info)

    def finish(self):
        """Completes `async_launch` evaluations on the current stream and raises their error, if any."""
        info = _capi.OobInfo()
        st = _capi.lib().ndi_interp1d_finish(self._h, current_stream_ptr(self._device), C.byref(info))
        self._inflight.clear()
        if st != _capi.OK:
            raise_eval(st, info)

    def trim(self):
        """Frees the handle's idle scratch sets and a library-owned ring (ndi_interp1d_trim)."""
        _capi.lib().ndi_interp1d_trim(self._h)

    def interp_array_ring(self, xs_flat, chunk_queries, consumer=None, *, slots=None, n_slots=2):
        """Chunked evaluation through a device-output ring (ndi_interp1d_eval_ring): `interp_array` for batches
        whose whole output does not fit (or need not stay) in device memory.  `slots`: list of contiguous device
        tensors of shape (chunk_queries, lanes) -- the ring; None lets the library own `n_slots` buffers.
        `consumer(chunk, view)` is called once per chunk, in order, after the chunk's kernels are enqueued on the
        current stream; `view` is the slot tensor cut to the chunk's rows (None with a library-owned ring) and
        `chunk` the ndi_ring_chunk fields (index, q_begin, q_count, out, row_stride, slot, stream).  Work the
        consumer enqueues on the current stream is ordered before the slot's reuse; if it uses another stream it
        returns a `torch.cuda.Event` recorded there."""
        qb = Buf(xs_flat, self._np_dtype)
        ring = _capi.RingDesc()
        ring.chunk_queries = int(chunk_queries)
        keep_events = []
        if slots is not None:
            for t in slots:
                _check_out_dtype(t, self._np_dtype)
                if not (is_torch(t) and t.is_cuda and t.dim() == 2 and t.shape[1] == self._lanes
                        and t.shape[0] >= chunk_queries and (t.stride(1) == 1 or self._lanes == 1)
                        and t.stride(0) == slots[0].stride(0)):
                    raise TypeError("ring slots must be device tensors of shape (>= chunk_queries, lanes) with "
                                    "contiguous rows and one common row pitch (see striped_ring)")
            arr = (C.c_void_p * len(slots))(*[t.data_ptr() for t in slots])
            ring.slots = C.cast(arr, C.POINTER(C.c_void_p))
            ring.n_slots = len(slots)
            ring.row_stride = max(slots[0].stride(0), self._lanes)
        else:
            ring.n_slots = int(n_slots)
            ring.row_stride = self._lanes

        failed = []

        def _cb(_user, cptr):
            # an exception must not unwind through the C frames: remember the first one, stop consuming, re-raise
            # after the library call has returned
            if failed:
                return None
            try:
                c = cptr.contents
                view = slots[c.slot][:c.q_count] if slots is not None else None
                ev = consumer(c, view)
            except BaseException as e:  # noqa: BLE001
                failed.append(e)
                return None
            if ev is None:
                return None
            keep_events.append(ev)
            return ev.cuda_event
        cb = _capi.RING_CONSUMER(_cb) if consumer is not None else C.cast(None, _capi.RING_CONSUMER)
        opts = _capi.EvalOpts()
        opts.q_memspace = qb.memspace
        opts.out_memspace = _capi.MEM_DEVICE
        opts.path = self.path
        opts.stream = current_stream_ptr(self._device)
        info = _capi.OobInfo()
        st = _capi.lib().ndi_interp1d_eval_ring(self._h, qb.ptr, qb.size, C.byref(ring), cb, None,
                                                C.byref(opts), C.byref(info))
        del keep_events
        if failed:
            raise failed[0]
        if st != _capi.OK:
            raise_eval(st, info)

    def interp_into(self, interpolator, target, x):
        # single query through the same device path (Q = 1)
        out = np.empty((1, self._lanes), dtype=self._np_dtype)
        self.interp_array_into(interpolator, np.array([x], dtype=self._np_dtype), out)
        target[...] = out.reshape(target.shape)


def _check_out_dtype(out, dt):
    """The kernels write sizeof(data element) per output element: the buffer must have the data's element type
    (the reference enforces this at compile time)."""
    got = np_dtype_of(out)
    if got != np.dtype(dt):
        raise TypeError(f"output buffer has element type {got}, the interpolator's data is {np.dtype(dt)}")


def _default_device() -> int:
    import os
    return int(os.environ.get("LOCAL_RANK", "0")) if _capi.lib().ndi_device_count() > 1 else 0


def _to_device(a, device):
    import torch
    return a.to(device) if is_torch(a) else torch.as_tensor(np.asarray(a), device=device)


class Linear(Interp1DStrategyBuilder, _DeviceStrategy1D):
    """Linear Interpolation Strategy (src/interp1d/strategies/linear.rs)."""

    MINIMUM_DATA_LENGHT = 2  # linear.rs:52
    _kind = _capi.LINEAR

    def __init__(self):
        _DeviceStrategy1D.__init__(self)
        self._extrapolate = False
        self._device_req = None

    @staticmethod
    def new() -> "Linear":
        return Linear()

    def device(self, ordinal: int) -> "Linear":
        """Build-side option of this mirror (device selection): the HIP device that holds the tables.
        Default: the device of the data tensor, else LOCAL_RANK / device 0."""
        self._device_req = int(ordinal)
        return self

    def extrapolate(self, extrapolate: bool) -> "Linear":
        """does the strategy extrapolate? Default is `false` (linear.rs:23-26)"""
        self._extrapolate = bool(extrapolate)
        return self

    def build(self, x, data):
        if np_dtype_of(data) not in (np.dtype(np.float32), np.dtype(np.float64)):
            # integer (and other non-f32/f64) element types: the reference's generic per-query path
            from .generic_host import HostLinear
            return HostLinear(_host(x), _host(data), self._extrapolate)
        return self._create(x, data, extrapolate=self._extrapolate, device=self._device_req)


# ---- boundary conditions (cubic_spline.rs:153-217) ---------------------------------------------
class SingleBoundary:
    """Boundary condition for a single boundary (one side of one data row)."""

    def __init__(self, kind, value=0.0):
        self.kind, self.value = kind, float(value)

    def __eq__(self, o):
        return isinstance(o, SingleBoundary) and (self.kind, self.value) == (o.kind, o.value)

    @staticmethod
    def FirstDeriv(v):
        return SingleBoundary(_capi.BC_FIRST_DERIV, v)

    @staticmethod
    def SecondDeriv(v):
        return SingleBoundary(_capi.BC_SECOND_DERIV, v)


SingleBoundary.NotAKnot = SingleBoundary(_capi.BC_NOT_A_KNOT)
SingleBoundary.Natural = SingleBoundary(_capi.BC_NATURAL)
SingleBoundary.Clamped = SingleBoundary(_capi.BC_CLAMPED)


class RowBoundary:
    """Boundary condition for a single data row."""

    def __init__(self, left: SingleBoundary, right: SingleBoundary):
        self.left, self.right = left, right

    def __eq__(self, o):
        return isinstance(o, RowBoundary) and self.left == o.left and self.right == o.right

    @staticmethod
    def Mixed(left: SingleBoundary, right: SingleBoundary):
        return RowBoundary(left, right)


RowBoundary.NotAKnot = RowBoundary(SingleBoundary.NotAKnot, SingleBoundary.NotAKnot)
RowBoundary.Natural = RowBoundary(SingleBoundary.Natural, SingleBoundary.Natural)
RowBoundary.Clamped = RowBoundary(SingleBoundary.Clamped, SingleBoundary.Clamped)


class BoundaryCondition:
    """Boundary conditions for the whole dataset."""

    def __init__(self, tag, rows=None):
        self.tag, self.rows = tag, rows

    @staticmethod
    def Individual(rows):
        """Set individual boundary conditions for each row in the data: an array of `RowBoundary` of
        shape `[1, data.shape[1:]...]` (cubic_spline.rs:165-167, 332-340)."""
        return BoundaryCondition("Individual", np.asarray(rows, dtype=object))


BoundaryCondition.NotAKnot = BoundaryCondition("NotAKnot")
BoundaryCondition.Natural = BoundaryCondition("Natural")
BoundaryCondition.Clamped = BoundaryCondition("Clamped")
BoundaryCondition.Periodic = BoundaryCondition("Periodic")


class CubicSpline(Interp1DStrategyBuilder):
    """The CubicSpline 1d interpolation Strategy (Builder) -- cubic_spline.rs:85-88, 723-771."""

    MINIMUM_DATA_LENGHT = 3  # cubic_spline.rs:751

    def __init__(self):
        self._extrapolate = False
        self._boundary = BoundaryCondition.NotAKnot  # default, cubic_spline.rs:724-729
        self._device_req = None
        self._reference_order = False

    @staticmethod
    def new() -> "CubicSpline":
        return CubicSpline()

    def device(self, ordinal: int) -> "CubicSpline":
        """Build-side option of this mirror: the HIP device that holds the tables (see Linear.device)."""
        self._device_req = int(ordinal)
        return self

    def extrapolate(self, extrapolate: bool) -> "CubicSpline":
        self._extrapolate = bool(extrapolate)
        return self

    def reference_order(self, yes: bool = True) -> "CubicSpline":
        """Build-side option of this mirror (ndi_build_flags, NDI_BUILD_REFERENCE_ORDER): never re-associate the Thomas
        sweeps, so the a / b tables are bit-identical to CubicSpline::build's (cubic_spline.rs:678-721) for every shape.
        Default: narrow trailing axes on many knots (n >= 2048, lanes <= 256) take blocked sweeps that agree with the
        reference to a few ulp of the neighbouring entries (include/ndinterp.h states the bound)."""
        self._reference_order = bool(yes)
        return self

    def boundary(self, boundary: BoundaryCondition) -> "CubicSpline":
        self._boundary = boundary
        return self

    def build(self, x, data) -> "CubicSplineStrategy":
        bc = self._boundary
        strat = CubicSplineStrategy()
        kw = dict(extrapolate=self._extrapolate, device=self._device_req,
                  build_flags=_capi.BUILD_REFERENCE_ORDER if self._reference_order else 0)
        if bc.tag == "Periodic":
            kw["periodic"] = True
        elif bc.tag == "Individual":
            shape = tuple(data.shape)
            expect = (1,) + shape[1:]
            if tuple(bc.rows.shape) != expect:  # cubic_spline.rs:333-340
                raise BuilderError.ShapeError(
                    f"Boundary conditions array has wrong shape. Expected: {list(expect)}, got: {list(bc.rows.shape)}")
            rows = bc.rows.reshape(-1)
            kw["per_lane"] = ([r.left.kind for r in rows], [r.left.value for r in rows],
                              [r.right.kind for r in rows], [r.right.value for r in rows])
            if all(r == rows[0] for r in rows):  # identical rows are one global boundary
                kw.pop("per_lane")
                kw["left"] = (rows[0].left.kind, rows[0].left.value)
                kw["right"] = (rows[0].right.kind, rows[0].right.value)
        else:
            k = {"NotAKnot": _capi.BC_NOT_A_KNOT, "Natural": _capi.BC_NATURAL, "Clamped": _capi.BC_CLAMPED}[bc.tag]
            kw["left"] = kw["right"] = (k, 0.0)
        try:
            return strat._create(x, data, **kw)
        except BuilderError.ValueError as e:
            raise BuilderError.ValueError(_periodic_mismatch_message(data)) from e


def _rust_debug_row(row: np.ndarray) -> str:
    """`{:?}` of an ndarray row view, as ndarray 0.17 prints a small 1-D array."""
    from .errors import _rust_float
    body = ", ".join(_rust_float(float(v)) for v in row.reshape(-1))
    if row.ndim == 1:
        return f"[{body}], shape=[{row.shape[0]}], strides=[1], layout=CFcf (0xf), const ndim=1"
    return f"[{body}], shape={list(row.shape)}"


def _periodic_mismatch_message(data) -> str:
    """cubic_spline.rs:483-488 / 501-506: the two message forms of the periodic ValueError."""
    from .errors import _rust_float
    d = _host(data)
    head = "for periodic boundary condition the first and last value must be equal. "
    if d.ndim == 1:
        return head + f"First: {_rust_float(float(d[0]))}, last: {_rust_float(float(d[-1]))}"
    return head + f"First: {_rust_debug_row(d[0])}, last: {_rust_debug_row(d[-1])}"


class CubicSplineStrategy(_DeviceStrategy1D):
    """The CubicSpline 1d interpolation Strategy (Implementation) -- `a`, `b` live on the device
    (cubic_spline.rs:94-102)."""

    _kind = _capi.CUBIC_SPLINE

    def coefficients(self):
        """Copies the tables `a`, `b` (each `(n-1, lanes)`) to the host."""
        n = self._n
        a = np.empty((n - 1, self._lanes), dtype=self._np_dtype)
        b = np.empty_like(a)
        st = _capi.lib().ndi_interp1d_coefficients(self._h, a.ctypes.data, b.ctypes.data, _capi.MEM_HOST)
        if st != _capi.OK:
            raise_builder(st)
        return a, b

    def _create(self, x, data, **kw):
        self._n = data.shape[0]
        return super()._create(x, data, **kw)


# ------------------------------------------------------------------------------------------------
# Interp1D / Interp1DBuilder (src/interp1d/mod.rs)
# ------------------------------------------------------------------------------------------------
def _host(a) -> np.ndarray:
    return a.detach().cpu().numpy() if is_torch(a) else np.asarray(a)


class Interp1D:
    """One dimensional interpolator (interp1d/mod.rs:39-51)."""

    def __init__(self, x, data, strategy):
        self.x, self.data, self.strategy = x, data, strategy
        self._x_host = _host(x)

    # construction ---------------------------------------------------------------------------
    @staticmethod
    def builder(data) -> "Interp1DBuilder":
        return Interp1DBuilder.new(data)

    @staticmethod
    def new_unchecked(x, data, strategy) -> "Interp1D":
        """Create a interpolator without any data validation (interp1d/mod.rs:363-365)."""
        return Interp1D(x, data, strategy)

    # helpers the strategies use ------------------------------------------------------------------
    def index_point(self, index: int):
        return self._x_host[index], self.data[index]

    def get_index_left_of(self, x) -> int:
        k = self._x_host
        if k.dtype in (np.float32, np.float64):
            r = int(get_lower_index(np.ascontiguousarray(k), np.array([x], dtype=k.dtype))[0])
            if r < 0:
                raise Panic("not implemented: failed to convert NaN to usize")
            return r
        return int(np.clip(np.searchsorted(k, x, side="right") - 1, 0, k.size - 2))

    def is_in_range(self, x) -> bool:
        return bool(self._x_host[0] <= x <= self._x_host[-1])

    # queries ----------------------------------------------------------------------------------
    def _lanes_shape(self):
        return tuple(self.data.shape[1:])

    def interp_scalar(self, x):
        """interp1d/mod.rs:108-114 (data must be 1-D)."""
        if len(self.data.shape) != 1:
            raise TypeError("interp_scalar needs 1-D data; use interp()")
        buf = np.zeros((), dtype=np_dtype_of(self.data))
        self.strategy.interp_into(self, buf, x)
        return buf[()]

    def interp(self, x):
        """interp1d/mod.rs:150-156."""
        target = np.zeros(self._lanes_shape(), dtype=np_dtype_of(self.data))
        self.strategy.interp_into(self, target, x)
        return target

    def interp_into(self, x, buffer):
        """interp1d/mod.rs:169-175; panics on a wrong buffer shape."""
        if tuple(buffer.shape) != self._lanes_shape():
            raise Panic(f"ShapeError/IncompatibleShape: incompatible shapes expected: "
                        f"{list(self._lanes_shape())}, got: {list(buffer.shape)}")
        self.strategy.interp_into(self, buffer, x)

    def get_buffer_shape(self, q_shape):
        """interp1d/mod.rs:346-354: query dims chained with data.shape[1..]."""
        return tuple(q_shape) + self._lanes_shape()

    def interp_array(self, xs):
        """interp1d/mod.rs:197-211.  numpy in -> numpy out; torch device tensor in -> device tensor out."""
        shape = self.get_buffer_shape(tuple(xs.shape))
        if is_torch(xs) and xs.is_cuda:
            import torch
            # element type of the *data* (the kernels write that; queries are converted to it)
            tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}.get(
                np_dtype_of(self.data))
            if tdt is None:
                raise TypeError("device query tensors need f32 / f64 data; other element types use host arrays")
            nbytes = int(np.prod(shape, dtype=np.int64)) * np_dtype_of(self.data).itemsize
            if nbytes >= OUTPUT_OWNED_MIN_BYTES:     # Array::zeros through the library's placement-checked allocator
                ys = output_empty(shape, np_dtype_of(self.data), xs.device.index or 0)
            else:
                ys = torch.empty(shape, dtype=tdt, device=xs.device)
            # the reference hands back zeros for rows it never reached only on Err, where the buffer
            # is dropped anyway (:210); no memset of the output is needed
        else:
            ys = np.zeros(shape, dtype=np_dtype_of(self.data))
        # the buffer is this call's own and is dropped on Err (:210): strategies that can use the knowledge are told
        self.interp_array_into(xs, ys, **({"fresh": True} if getattr(self.strategy, "_takes_fresh", False) else {}))
        return ys

    def interp_array_into(self, xs, buffer, **kw):
        """interp1d/mod.rs:272-324.  Any query rank: a flatten of the query array and a 2-D view of
        the buffer (rows = queries, columns = lanes); panics when the buffer shape is wrong."""
        expect = self.get_buffer_shape(tuple(xs.shape))
        if tuple(buffer.shape) != expect:
            raise Panic(f"ShapeError/IncompatibleShape: incompatible shapes expected: {list(expect)}, "
                        f"got: {list(buffer.shape)}")
        if np_dtype_of(buffer) != np_dtype_of(self.data):
            raise TypeError(f"buffer has element type {np_dtype_of(buffer)}, the data is {np_dtype_of(self.data)}")
        nq = int(np.prod(xs.shape, dtype=np.int64))
        lanes = int(np.prod(self._lanes_shape(), dtype=np.int64))
        xs_flat = xs.reshape(-1)
        if is_torch(buffer):
            if not buffer.is_contiguous():
                raise TypeError("device output buffers must be contiguous")
            out2d = buffer.view(nq, lanes)
            self.strategy.interp_array_into(self, xs_flat, out2d, **kw)
            return
        if buffer.flags.c_contiguous:
            self.strategy.interp_array_into(self, _host(xs_flat) if not is_torch(xs) else xs_flat,
                                            buffer.reshape(nq, lanes), **kw)
            return
        # strided ArrayViewMut: bounce through a contiguous temporary
        tmp = np.zeros((nq, lanes), dtype=np_dtype_of(self.data))
        done = nq
        try:
            self.strategy.interp_array_into(self, _host(xs_flat), tmp, **kw)
        except (InterpolateError.OutOfBounds, Panic) as e:
            done = e.index if getattr(e, "index", None) is not None else 0   # rows before the failing query are
            raise                                          # written, later rows stay untouched (interp1d/mod.rs:334-342)
        except BaseException:
            done = 0                                       # device failure: nothing in tmp can be trusted
            raise
        finally:
            if done and len(xs.shape) == 0:
                buffer[...] = tmp[0].reshape(buffer.shape)
            elif done:
                where = np.unravel_index(np.arange(done), tuple(xs.shape))     # works for any strides
                buffer[where] = tmp[:done].reshape((done,) + self._lanes_shape())

    def replicate(self, devices):
        """Replicas of this interpolator on the given devices (tables copied device to device, nothing rebuilt):
        what `sharding.interp_array_sharded` takes.  `interp.replicate(range(pkg.device_count()))`."""
        if not hasattr(self.strategy, "clone"):
            raise TypeError("replicate needs a built-in device strategy (f32 / f64 data)")
        return [Interp1D(self.x, self.data, self.strategy.clone(d)) for d in devices]

    def interp_array_ring(self, xs, chunk_queries, consumer=None, *, slots=None, n_slots=2):
        """`interp_array` (interp1d/mod.rs:197-211) for outputs larger than device memory: the flattened
        queries are evaluated `chunk_queries` rows at a time into a ring of device buffers and handed to
        `consumer(chunk, rows)`; see `_DeviceStrategy1D.interp_array_ring` / ndi_interp1d_eval_ring."""
        if not hasattr(self.strategy, "interp_array_ring"):
            raise TypeError("the ring evaluation needs a built-in device strategy (f32 / f64 data)")
        self.strategy.interp_array_ring(xs.reshape(-1), chunk_queries, consumer, slots=slots, n_slots=n_slots)


class Interp1DBuilder:
    """Create and configure a `Interp1D` interpolator (interp1d/mod.rs:60-70, 389-477)."""

    def __init__(self, data, x=None, strategy=None):
        self._data = data
        self._x = x
        self._strategy = strategy if strategy is not None else Linear.new()  # :408

    @staticmethod
    def new(data) -> "Interp1DBuilder":
        return Interp1DBuilder(data)

    def x(self, x) -> "Interp1DBuilder":
        return Interp1DBuilder(self._data, x, self._strategy)

    def strategy(self, strategy) -> "Interp1DBuilder":
        return Interp1DBuilder(self._data, self._x, strategy)

    def build(self) -> Interp1D:
        """Validate input data and create the configured `Interp1D` (interp1d/mod.rs:443-476)."""
        data, strategy = self._data, self._strategy
        shape = tuple(data.shape)
        if len(shape) < 1:
            raise BuilderError.ShapeError("data dimension is 0, needs to be at least 1")
        n = shape[0]
        dt = np_dtype_of(data)
        if self._x is None:  # default axis 0..len cast to T (:402-406)
            x = np.arange(n).astype(dt)
        else:
            x = self._x
        need = type(strategy).MINIMUM_DATA_LENGHT
        if n < need:
            raise BuilderError.NotEnoughData(
                f"The chosen Interpolation strategy needs at least {need} data points")
        if monotonic_prop(x) != Monotonic.Rising(True):
            raise BuilderError.Monotonic("Values in the x axis need to be strictly monotonic rising")
        x_len = int(np.prod(x.shape, dtype=np.int64))
        if x_len != n:
            raise BuilderError.ShapeError(
                f"Lengths of x and data axis need to match. Got x: {x_len}, data: {n}")
        finished = strategy.build(x, data)
        return Interp1D(x, data, finished)
