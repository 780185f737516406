"""Host-side mirror of the reference's 2-D surface (src/interp2d/mod.rs + strategies/):
`Interp2DBuilder`, `Interp2D`, the strategy trait pair and the built-in `Bilinear`."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._arrays import OUTPUT_OWNED_MIN_BYTES, Buf, current_stream_ptr, dtype_id, is_torch, np_dtype_of, output_empty
from .errors import BuilderError, InterpolateError, Panic, raise_builder, raise_eval
from .interp1d import _check_out_dtype, _default_device, _host, _to_device
from .vector_extensions import Monotonic, get_lower_index, monotonic_prop


class Interp2DStrategyBuilder:
    """Trait `Interp2DStrategyBuilder` (src/interp2d/strategies/mod.rs:14-44)."""

    MINIMUM_DATA_LENGHT = 2

    def build(self, x, y, data) -> "Interp2DStrategy":
        raise NotImplementedError


class Interp2DStrategy:
    """Trait `Interp2DStrategy` (src/interp2d/strategies/mod.rs:46-73): per-query
    `interp_into(interpolator, target, x, y)`; `interp_array_into` is the defaulted batched hook whose
    default body is the reference's serial loop (interp2d/mod.rs:287-307)."""

    def interp_into(self, interpolator, target, x, y) -> None:
        raise NotImplementedError

    def interp_array_into(self, interpolator, xs_flat, ys_flat, out2d) -> None:
        shape = tuple(interpolator.data.shape[2:])
        for i in range(len(xs_flat)):
            self.interp_into(interpolator, out2d[i].reshape(shape), xs_flat[i], ys_flat[i])

    def release(self) -> None:
        pass


class Bilinear(Interp2DStrategyBuilder, Interp2DStrategy):
    """Bilinear strategy (src/interp2d/strategies/bilinear.rs); builder and finished strategy in one,
    as in the reference (`type FinishedStrat = Self`, :43)."""

    MINIMUM_DATA_LENGHT = 2  # bilinear.rs:41
    path = _capi.PATH_AUTO   # evaluation formulation (ndi_path): BUCKETED = tile-grouped query order

    def __init__(self):
        self._extrapolate = False
        self._h = None
        self._device = 0
        self._device_req = None
        self._np_dtype = None
        self._lanes = 1
        self._inflight = []

    @staticmethod
    def new() -> "Bilinear":
        return Bilinear()

    def device(self, ordinal: int) -> "Bilinear":
        """Build-side option of this mirror: the HIP device that holds the grid (default: the data tensor's
        device, else LOCAL_RANK / device 0)."""
        self._device_req = int(ordinal)
        return self

    def extrapolate(self, yes: bool) -> "Bilinear":
        self._extrapolate = bool(yes)
        return self

    def build(self, x, y, data, device=None):
        dt = np_dtype_of(data)
        if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
            # integer (and other non-f32/f64) element types: the reference's generic per-query path
            from .generic_host import HostBilinear
            return HostBilinear(_host(x), _host(y), _host(data), self._extrapolate)
        tid = dtype_id(dt)
        db = Buf(data)

        def axis(a):
            if a is None:
                return None
            if db.memspace == _capi.MEM_HOST:
                return Buf(_host(a), dt)
            return Buf(_to_device(a, db.keep.device), dt)

        xb, yb = axis(x), axis(y)
        if device is None:
            device = self._device_req
        if device is None:
            device = db.device if db.memspace == _capi.MEM_DEVICE else _default_device()
        nx, ny = db.shape[0], db.shape[1]
        lanes = int(np.prod(db.shape[2:], dtype=np.int64)) if len(db.shape) > 2 else 1
        d = _capi.Interp2DDesc()
        d.dtype, d.extrapolate, d.device, d.memspace = tid, int(self._extrapolate), device, db.memspace
        d.nx, d.ny, d.lanes = nx, ny, lanes
        d.x_len = xb.size if xb is not None else nx
        d.y_len = yb.size if yb is not None else ny
        d.x = xb.ptr if xb is not None else None
        d.y = yb.ptr if yb is not None else None
        d.data = db.ptr
        d.validate = 0  # Interp2DBuilder.build() validated already (interp2d/mod.rs:477-511)
        h = C.c_void_p()
        st = _capi.lib().ndi_interp2d_create(C.byref(d), C.byref(h))
        if st != _capi.OK:
            raise_builder(st)
        self._h, self._device, self._np_dtype, self._lanes = h, device, dt, lanes
        return self

    def release(self):
        if self._h is not None:
            _capi.lib().ndi_interp2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def clone(self, device: int):
        """A replica of this built strategy on `device` (ndi_interp2d_clone): the grid is copied device to device."""
        import copy
        h = C.c_void_p()
        st = _capi.lib().ndi_interp2d_clone(self._h, int(device), C.byref(h))
        if st != _capi.OK:
            raise_builder(st)
        other = copy.copy(self)
        other._h, other._device, other._inflight = h, int(device), []
        return other

    _takes_fresh = True   # see _DeviceStrategy1D

    def interp_array_into(self, interpolator, xs_flat, ys_flat, out2d, *, async_launch=False, fresh=False,
                          rows_after_error_unspecified=False):
        """Replaces the reference's query loop (interp2d/mod.rs:287-307) by one C-ABI call.  `fresh`: the buffer was
        allocated for this call and is dropped on Err (Interp2D::interp_array, :175-196) -- NDI_EVAL_FRESH_OUTPUT."""
        qx = Buf(xs_flat, self._np_dtype)
        qy = Buf(ys_flat, self._np_dtype)
        if qx.memspace != qy.memspace:
            raise TypeError("xs and ys must live in the same memory space")
        _check_out_dtype(out2d, self._np_dtype)
        if async_launch:                      # every async batch reads its query arrays until finish():
            self._inflight.append((qx, qy))   # keep all (possibly converted) copies alive, not only the last
        opts = _capi.EvalOpts()
        opts.q_memspace = qx.memspace
        opts.path = self.path
        opts.async_launch = int(bool(async_launch))
        # rows_after_error_unspecified: a caller-owned buffer whose rows at / after a failing query the caller gives up
        # (NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED; the reference leaves them untouched, interp1d/mod.rs:334-342)
        opts.flags = (_capi.EVAL_FRESH_OUTPUT if fresh else _capi.EVAL_DEFAULT) | \
            (_capi.EVAL_ROWS_AFTER_ERROR_UNSPECIFIED if rows_after_error_unspecified else 0)
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
            if qx.memspace == _capi.MEM_DEVICE:
                opts.stream = current_stream_ptr(self._device)
        info = _capi.OobInfo()
        st = _capi.lib().ndi_interp2d_eval(self._h, qx.ptr, qy.ptr, qx.size, optr, max(stride, self._lanes),
                                           C.byref(opts), C.byref(info))
        if st != _capi.OK:
            raise_eval(st, info)

    def finish(self):
        info = _capi.OobInfo()
        st = _capi.lib().ndi_interp2d_finish(self._h, current_stream_ptr(self._device), C.byref(info))
        self._inflight.clear()
        if st != _capi.OK:
            raise_eval(st, info)

    def trim(self):
        _capi.lib().ndi_interp2d_trim(self._h)

    def probe_ceiling(self, out2d, reps=5) -> float:
        """ms of the evaluation kernel's memory access mix alone on this handle's grid (ndi_interp2d_probe_ceiling);
        `out2d`: a device tensor (nq, lanes) that is overwritten."""
        ms = C.c_double()
        st = _capi.lib().ndi_interp2d_probe_ceiling(self._h, out2d.shape[0], out2d.data_ptr(), out2d.stride(0),
                                                    current_stream_ptr(self._device), int(reps), C.byref(ms))
        if st != _capi.OK:
            from .errors import DeviceError
            raise DeviceError(_capi.last_error())
        return ms.value

    def interp_array_ring(self, xs_flat, ys_flat, chunk_queries, consumer=None, *, slots=None, n_slots=2):
        """ndi_interp2d_eval_ring; see `_DeviceStrategy1D.interp_array_ring`."""
        qx, qy = Buf(xs_flat, self._np_dtype), Buf(ys_flat, self._np_dtype)
        if qx.memspace != qy.memspace:
            raise TypeError("xs and ys must live in the same memory space")
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
        opts.q_memspace = qx.memspace
        opts.out_memspace = _capi.MEM_DEVICE
        opts.path = self.path
        opts.stream = current_stream_ptr(self._device)
        info = _capi.OobInfo()
        st = _capi.lib().ndi_interp2d_eval_ring(self._h, qx.ptr, qy.ptr, qx.size, C.byref(ring), cb, None,
                                                C.byref(opts), C.byref(info))
        del keep_events
        if failed:
            raise failed[0]
        if st != _capi.OK:
            raise_eval(st, info)

    def interp_into(self, interpolator, target, x, y):
        out = np.empty((1, self._lanes), dtype=self._np_dtype)
        self.interp_array_into(interpolator, np.array([x], dtype=self._np_dtype),
                               np.array([y], dtype=self._np_dtype), out)
        target[...] = out.reshape(target.shape)


class Interp2D:
    """Two dimensional interpolator (interp2d/mod.rs:36-48)."""

    def __init__(self, x, y, data, strategy):
        self.x, self.y, self.data, self.strategy = x, y, data, strategy
        self._x_host, self._y_host = _host(x), _host(y)

    @staticmethod
    def builder(data) -> "Interp2DBuilder":
        return Interp2DBuilder.new(data)

    @staticmethod
    def new_unchecked(x, y, data, strategy) -> "Interp2D":
        return Interp2D(x, y, data, strategy)

    def index_point(self, x_idx: int, y_idx: int):
        return self._x_host[x_idx], self._y_host[y_idx], self.data[x_idx, y_idx]

    def get_index_left_of(self, x, y):
        def one(k, v):
            if k.dtype in (np.float32, np.float64):
                r = int(get_lower_index(np.ascontiguousarray(k), np.array([v], dtype=k.dtype))[0])
                if r < 0:
                    raise Panic("not implemented: failed to convert NaN to usize")
                return r
            return int(np.clip(np.searchsorted(k, v, side="right") - 1, 0, k.size - 2))
        return one(self._x_host, x), one(self._y_host, y)

    def is_in_x_range(self, x) -> bool:
        return bool(self._x_host[0] <= x <= self._x_host[-1])

    def is_in_y_range(self, y) -> bool:
        return bool(self._y_host[0] <= y <= self._y_host[-1])

    def _lanes_shape(self):
        return tuple(self.data.shape[2:])

    def interp_scalar(self, x, y):
        """interp2d/mod.rs:107-113 (data must be 2-D)."""
        if len(self.data.shape) != 2:
            raise TypeError("interp_scalar needs 2-D data; use interp()")
        buf = np.zeros((), dtype=np_dtype_of(self.data))
        self.strategy.interp_into(self, buf, x, y)
        return buf[()]

    def interp(self, x, y):
        target = np.zeros(self._lanes_shape(), dtype=np_dtype_of(self.data))
        self.strategy.interp_into(self, target, x, y)
        return target

    def interp_into(self, x, y, buffer):
        if tuple(buffer.shape) != self._lanes_shape():
            raise Panic(f"ShapeError/IncompatibleShape: incompatible shapes expected: "
                        f"{list(self._lanes_shape())}, got: {list(buffer.shape)}")
        self.strategy.interp_into(self, buffer, x, y)

    def get_buffer_shape(self, q_shape):
        """interp2d/mod.rs:310-321."""
        return tuple(q_shape) + self._lanes_shape()

    def interp_array(self, xs, ys):
        """interp2d/mod.rs:175-196; panics when `xs.shape != ys.shape`."""
        if tuple(xs.shape) != tuple(ys.shape):
            raise Panic("`xs.shape()` and `ys.shape()` do not match")
        shape = self.get_buffer_shape(tuple(xs.shape))
        if is_torch(xs) and xs.is_cuda:
            import torch
            tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}.get(
                np_dtype_of(self.data))
            if tdt is None:
                raise TypeError("device query tensors need f32 / f64 data; other element types use host arrays")
            nbytes = int(np.prod(shape, dtype=np.int64)) * np_dtype_of(self.data).itemsize
            if nbytes >= OUTPUT_OWNED_MIN_BYTES:     # Array::zeros through the library's placement-checked allocator
                zs = output_empty(shape, np_dtype_of(self.data), xs.device.index or 0)
            else:
                zs = torch.empty(shape, dtype=tdt, device=xs.device)
        else:
            zs = np.zeros(shape, dtype=np_dtype_of(self.data))
        # the buffer is this call's own and is dropped on Err (:193-195): strategies that can use the knowledge are told
        self.interp_array_into(xs, ys, zs, **({"fresh": True} if getattr(self.strategy, "_takes_fresh", False) else {}))
        return zs

    def interp_array_into(self, xs, ys, buffer, **kw):
        """interp2d/mod.rs:215-285."""
        if tuple(xs.shape) != tuple(ys.shape):
            raise Panic("`xs.shape()` and `ys.shape()` do not match")
        expect = self.get_buffer_shape(tuple(xs.shape))
        if tuple(buffer.shape) != expect:
            raise Panic(f"ShapeError/IncompatibleShape: incompatible shapes expected: {list(expect)}, "
                        f"got: {list(buffer.shape)}")
        if np_dtype_of(buffer) != np_dtype_of(self.data):
            raise TypeError(f"buffer has element type {np_dtype_of(buffer)}, the data is {np_dtype_of(self.data)}")
        nq = int(np.prod(xs.shape, dtype=np.int64))
        lanes = int(np.prod(self._lanes_shape(), dtype=np.int64))
        xf, yf = xs.reshape(-1), ys.reshape(-1)
        if is_torch(buffer):
            if not buffer.is_contiguous():
                raise TypeError("device output buffers must be contiguous")
            self.strategy.interp_array_into(self, xf, yf, buffer.view(nq, lanes), **kw)
            return
        if not is_torch(xs):
            xf, yf = _host(xf), _host(yf)
        if buffer.flags.c_contiguous:
            self.strategy.interp_array_into(self, xf, yf, buffer.reshape(nq, lanes), **kw)
            return
        tmp = np.zeros((nq, lanes), dtype=np_dtype_of(self.data))
        done = nq
        try:
            self.strategy.interp_array_into(self, xf, yf, tmp, **kw)
        except (InterpolateError.OutOfBounds, Panic) as e:
            done = e.index if getattr(e, "index", None) is not None else 0   # rows before the failing query are
            raise                                          # written, later rows stay untouched (interp2d/mod.rs:297-306)
        except BaseException:
            done = 0                                       # device failure: nothing in tmp can be trusted
            raise
        finally:
            if done and len(xs.shape) == 0:
                buffer[...] = tmp[0].reshape(buffer.shape)
            elif done:
                where = np.unravel_index(np.arange(done), tuple(xs.shape))
                buffer[where] = tmp[:done].reshape((done,) + self._lanes_shape())

    def replicate(self, devices):
        """Replicas of this interpolator on the given devices (see Interp1D.replicate)."""
        if not hasattr(self.strategy, "clone"):
            raise TypeError("replicate needs a built-in device strategy (f32 / f64 data)")
        return [Interp2D(self.x, self.y, self.data, self.strategy.clone(d)) for d in devices]

    def interp_array_ring(self, xs, ys, chunk_queries, consumer=None, *, slots=None, n_slots=2):
        """`interp_array` (interp2d/mod.rs:175-196) through a device-output ring (ndi_interp2d_eval_ring)."""
        if tuple(xs.shape) != tuple(ys.shape):
            raise Panic("`xs.shape()` and `ys.shape()` do not match")
        if not hasattr(self.strategy, "interp_array_ring"):
            raise TypeError("the ring evaluation needs the built-in device strategy (f32 / f64 data)")
        self.strategy.interp_array_ring(xs.reshape(-1), ys.reshape(-1), chunk_queries, consumer, slots=slots,
                                        n_slots=n_slots)


class Interp2DBuilder:
    """Create and configure a `Interp2D` interpolator (interp2d/mod.rs:52-64, 382-519)."""

    def __init__(self, data, x=None, y=None, strategy=None):
        self._data, self._x, self._y = data, x, y
        self._strategy = strategy if strategy is not None else Bilinear.new()  # :403

    @staticmethod
    def new(data) -> "Interp2DBuilder":
        return Interp2DBuilder(data)

    def strategy(self, strategy) -> "Interp2DBuilder":
        return Interp2DBuilder(self._data, self._x, self._y, strategy)

    def x(self, x) -> "Interp2DBuilder":
        return Interp2DBuilder(self._data, x, self._y, self._strategy)

    def y(self, y) -> "Interp2DBuilder":
        return Interp2DBuilder(self._data, self._x, y, self._strategy)

    def build(self) -> Interp2D:
        """Validate the input and create the configured `Interp2D` (interp2d/mod.rs:468-518)."""
        data, strategy = self._data, self._strategy
        shape = tuple(data.shape)
        if len(shape) < 2:
            raise BuilderError.ShapeError("data dimension needs to be at least 2")
        need = type(strategy).MINIMUM_DATA_LENGHT
        if shape[0] < need:
            raise BuilderError.NotEnoughData(
                "The 0-dimension has not enough data for the chosen interpolation strategy. "
                f"Provided: {shape[0]}, Reqired: {need}")
        if shape[1] < need:
            raise BuilderError.NotEnoughData(
                "The 1-dimension has not enough data for the chosen interpolation strategy. "
                f"Provided: {shape[1]}, Reqired: {need}")
        dt = np_dtype_of(data)
        x = np.arange(shape[0]).astype(dt) if self._x is None else self._x
        y = np.arange(shape[1]).astype(dt) if self._y is None else self._y
        x_len = int(np.prod(x.shape, dtype=np.int64))
        y_len = int(np.prod(y.shape, dtype=np.int64))
        if x_len != shape[0]:
            raise BuilderError.ShapeError(
                f"Lenghts of x-axis and data-0-axis need to match. Got x: {x_len}, data-0: {shape[0]}")
        if y_len != shape[1]:
            raise BuilderError.ShapeError(
                f"Lenghts of y-axis and data-1-axis need to match. Got y: {y_len}, data-1: {shape[1]}")
        if monotonic_prop(x) != Monotonic.Rising(True):
            raise BuilderError.Monotonic("The x-axis needs to be strictly monotonic rising")
        if monotonic_prop(y) != Monotonic.Rising(True):
            raise BuilderError.Monotonic("The y-axis needs to be strictly monotonic rising")
        finished = strategy.build(x, y, data)
        return Interp2D(x, y, data, finished)
