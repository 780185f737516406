"""Error values of the reference (src/lib.rs:127-146) as Python exceptions.

`BuilderError.NotEnoughData` etc. read like the Rust enum variants; a Rust panic on the path
(wrong buffer shape, NaN query while extrapolating, mismatched query shapes) is a `Panic`."""
from __future__ import annotations

from . import _capi


class BuilderError(Exception):
    """Errors during Interpolator creation (src/lib.rs:127-139)."""


class _NotEnoughData(BuilderError):
    pass


class _Monotonic(BuilderError):
    pass


class _ShapeError(BuilderError):
    pass


class _ValueError(BuilderError):
    pass


BuilderError.NotEnoughData = _NotEnoughData
BuilderError.Monotonic = _Monotonic
BuilderError.ShapeError = _ShapeError
BuilderError.ValueError = _ValueError
for _c, _n in ((_NotEnoughData, "NotEnoughData"), (_Monotonic, "Monotonic"), (_ShapeError, "ShapeError"),
               (_ValueError, "ValueError")):
    _c.__name__ = _n
    _c.__qualname__ = "BuilderError." + _n


class InterpolateError(Exception):
    """Errors during Interpolation (src/lib.rs:142-146)."""


class _OutOfBounds(InterpolateError):
    def __init__(self, msg, index=None, value=None, axis=0):
        super().__init__(msg)
        self.index = index  # lowest failing flat query index (interp_array stops there)
        self.value = value
        self.axis = axis


_OutOfBounds.__name__ = "OutOfBounds"
_OutOfBounds.__qualname__ = "InterpolateError.OutOfBounds"
InterpolateError.OutOfBounds = _OutOfBounds


class Panic(RuntimeError):
    """A condition on which the reference panics instead of returning Err.  `index` is the flat query index
    the serial loop would have panicked at, when the batch path knows it (NaN query while extrapolating)."""

    def __init__(self, msg, index=None):
        super().__init__(msg)
        self.index = index


class DeviceError(RuntimeError):
    """HIP failure / missing device / unsupported configuration (ABI-only codes)."""


def raise_builder(status: int, msg: str | None = None):
    msg = msg if msg is not None else _capi.last_error()
    if status == _capi.NOT_ENOUGH_DATA:
        raise BuilderError.NotEnoughData(msg)
    if status == _capi.MONOTONIC:
        raise BuilderError.Monotonic(msg)
    if status == _capi.SHAPE:
        raise BuilderError.ShapeError(msg)
    if status == _capi.VALUE:
        raise BuilderError.ValueError(msg)
    raise DeviceError(f"{_capi.STATUS_NAMES[status] if 0 <= status < 10 else status}: {msg}")


def _rust_float(v: float) -> str:
    # `{x:#?}` of a float: shortest round-trip repr; inf/NaN spelled as Rust does
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "inf" if v > 0 else "-inf"
    return repr(float(v))


def raise_eval(status: int, info: "_capi.OobInfo"):
    if status == _capi.OUT_OF_BOUNDS:
        name = "x" if info.axis == 0 else "y"
        # linear.rs:81-83, cubic_spline.rs:799-801, bilinear.rs:72-79
        raise InterpolateError.OutOfBounds(f"{name} = {_rust_float(info.value)} is not in range",
                                           index=int(info.index), value=float(info.value), axis=int(info.axis))
    if status == _capi.NAN_QUERY:
        # vector_extensions.rs:83-84 (unimplemented! -> panic)
        raise Panic("not implemented: failed to convert NaN to usize", index=int(info.index))
    raise DeviceError(f"{_capi.STATUS_NAMES[status] if 0 <= status < 10 else status}: {_capi.last_error()}")
