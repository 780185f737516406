"""`VectorExtensions` of the reference (src/vector_extensions.rs): monotonic_prop and the batched
get_lower_index."""
from __future__ import annotations

import numpy as np

from . import _capi
from ._arrays import Buf, dtype_id, is_torch
from .errors import DeviceError


class Monotonic:
    """Monotonic enum, src/vector_extensions.rs:25-29."""

    def __init__(self, kind: str, strict: bool = False):
        self.kind, self.strict = kind, strict

    def __eq__(self, other):
        return isinstance(other, Monotonic) and (self.kind, self.strict if self.kind != "NotMonotonic" else False) == \
            (other.kind, other.strict if other.kind != "NotMonotonic" else False)

    def __repr__(self):
        return "NotMonotonic" if self.kind == "NotMonotonic" else f"{self.kind} {{ strict: {str(self.strict).lower()} }}"

    @staticmethod
    def Rising(strict):
        return Monotonic("Rising", strict)

    @staticmethod
    def Falling(strict):
        return Monotonic("Falling", strict)


Monotonic.NotMonotonic = Monotonic("NotMonotonic")
_FROM_CODE = {0: Monotonic.NotMonotonic, 1: Monotonic.Rising(True), 2: Monotonic.Rising(False),
              3: Monotonic.Falling(True), 4: Monotonic.Falling(False)}


def _monotonic_generic(v: np.ndarray) -> Monotonic:
    """Host-side scan for element types outside f32/f64 (e.g. the i32 axes of tests/interp1d.rs:123-140);
    same state machine as src/vector_extensions.rs:116-198."""
    if v.size <= 1:
        return Monotonic.NotMonotonic
    a, b = v[:-1], v[1:]
    lt, eq, gt = a < b, a == b, a > b
    strict = not bool(eq.any())
    noneq = ~eq
    if not noneq.any():
        return Monotonic.NotMonotonic
    first = int(np.argmax(noneq))
    rising = bool(lt[first])
    ok = (lt | eq) if rising else (gt | eq)
    if not bool(ok.all()):
        return Monotonic.NotMonotonic
    return Monotonic.Rising(strict) if rising else Monotonic.Falling(strict)


def monotonic_prop(v) -> Monotonic:
    """VectorExtensions::monotonic_prop (src/vector_extensions.rs:40-53)."""
    a = v.detach().cpu().numpy() if is_torch(v) else np.asarray(v)
    if a.dtype in (np.float32, np.float64):
        a = np.ascontiguousarray(a)
        return _FROM_CODE[_capi.lib().ndi_monotonic_prop(dtype_id(a.dtype), a.ctypes.data, a.size)]
    return _monotonic_generic(a)


def get_lower_index(knots, xs, device: int = 0):
    """VectorExtensions::get_lower_index (src/vector_extensions.rs:55-111) for a batch of queries, on the
    device (wavefront-cooperative search).  Returns int64 indices; -1 marks a NaN query (the reference
    panics there)."""
    kb = Buf(knots)
    qb = Buf(xs, kb.np_dtype)
    if kb.memspace != qb.memspace:
        raise ValueError("knots and queries must live in the same memory space")
    if kb.memspace == _capi.MEM_DEVICE:
        import torch
        out = torch.empty(qb.size, dtype=torch.int64, device=qb.keep.device)
        optr, device = out.data_ptr(), kb.device
    else:
        out = np.empty(qb.size, dtype=np.int64)
        optr = out.ctypes.data
    st = _capi.lib().ndi_get_lower_index_batch(dtype_id(kb.np_dtype), device, kb.ptr, kb.size, qb.ptr, qb.size,
                                              optr, kb.memspace)
    if st != _capi.OK:
        raise DeviceError(f"{_capi.STATUS_NAMES[st]}: {_capi.last_error()}")
    return out.reshape(qb.shape)


class Locator:
    """`get_lower_index` with the knot pyramid resident on the device (ndi_locator_*): what
    `Interp1D::get_index_left_of` (src/interp1d/mod.rs:380-382) is to a built interpolator -- no allocation or
    knot upload per call."""

    def __init__(self, knots, device: int | None = None):
        import ctypes as C
        kb = Buf(knots)
        self._np_dtype = kb.np_dtype
        self._memspace = kb.memspace
        self._device = kb.device if kb.memspace == _capi.MEM_DEVICE else (device or 0)
        h = C.c_void_p()
        st = _capi.lib().ndi_locator_create(dtype_id(kb.np_dtype), self._device, kb.ptr, kb.size, kb.memspace,
                                            C.byref(h))
        if st != _capi.OK:
            raise DeviceError(f"{_capi.STATUS_NAMES[st]}: {_capi.last_error()}")
        self._h = h

    def get_lower_index(self, xs):
        """int64 indices (the shape of `xs`); -1 marks a NaN query.  numpy in -> numpy out, device tensor in ->
        device tensor out (on the current stream)."""
        from ._arrays import current_stream_ptr
        qb = Buf(xs, self._np_dtype)
        if qb.memspace == _capi.MEM_DEVICE:
            import torch
            out = torch.empty(qb.size, dtype=torch.int64, device=qb.keep.device)
            optr, stream = out.data_ptr(), current_stream_ptr(self._device)
        else:
            out = np.empty(qb.size, dtype=np.int64)
            optr, stream = out.ctypes.data, None
        st = _capi.lib().ndi_locator_eval(self._h, qb.ptr, qb.size, optr, qb.memspace, stream)
        if st != _capi.OK:
            raise DeviceError(f"{_capi.STATUS_NAMES[st]}: {_capi.last_error()}")
        return out.reshape(qb.shape)

    def release(self):
        if self._h is not None:
            _capi.lib().ndi_locator_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
