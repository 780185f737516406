"""ndarray-interp_amd -- MI355X-native `interp_array` hot path of ndarray-interp.

Layout
  csrc/                 hand-written HIP kernels (gfx950) + the C ABI (include/ndinterp.h)
  libndinterp_hip.so    built in-tree by __graft_entry__.build() / csrc/Makefile
  interp1d, interp2d    host-side mirror of Interp1DBuilder / Interp2DBuilder + the Strategy traits
  vector_extensions     monotonic_prop / batched get_lower_index / Locator (resident knot pyramid)
  generic_host          the reference's generic per-query path for non-f32/f64 element types (integers)
  sharding              query sharding over the GPUs of a node (no collective on the data path)

The directory name carries a hyphen (fixed by the project layout); it is imported by path as
`ndarray_interp_amd` (see tests/conftest.py, bench.py, __graft_entry__.py).
"""
from . import _capi
from ._arrays import OUTPUT_OWNED_MIN_BYTES, output_empty, output_trim, output_zeros, striped_ring
from .errors import BuilderError, DeviceError, InterpolateError, Panic
from .interp1d import (BoundaryCondition, CubicSpline, CubicSplineStrategy, Interp1D, Interp1DBuilder,
                       Interp1DStrategy, Interp1DStrategyBuilder, Linear, RowBoundary, SingleBoundary)
from .interp2d import Bilinear, Interp2D, Interp2DBuilder, Interp2DStrategy, Interp2DStrategyBuilder
from .vector_extensions import Locator, Monotonic, get_lower_index, monotonic_prop
from . import sharding

PATH_AUTO, PATH_GATHER, PATH_BUCKETED = _capi.PATH_AUTO, _capi.PATH_GATHER, _capi.PATH_BUCKETED


def device_count() -> int:
    return _capi.lib().ndi_device_count()


def profile_enable(on: bool) -> None:
    _capi.lib().ndi_profile_enable(int(bool(on)))


def profile_read(reset: bool = True) -> dict:
    import ctypes
    p = _capi.Profile()
    st = _capi.lib().ndi_profile_read(ctypes.byref(p), int(bool(reset)))
    if st != _capi.OK:
        raise DeviceError(_capi.last_error())
    return {"eval_launches": p.eval_launches, "eval_ms": p.eval_ms, "locate_launches": p.locate_launches,
            "locate_ms": p.locate_ms, "group_launches": p.group_launches, "group_ms": p.group_ms,
            "last_path": _capi.PATH_NAMES.get(p.last_path, str(p.last_path))}


__all__ = [
    "BuilderError", "InterpolateError", "Panic", "DeviceError",
    "Interp1D", "Interp1DBuilder", "Interp1DStrategy", "Interp1DStrategyBuilder", "Linear", "CubicSpline",
    "CubicSplineStrategy", "BoundaryCondition", "RowBoundary", "SingleBoundary",
    "Interp2D", "Interp2DBuilder", "Interp2DStrategy", "Interp2DStrategyBuilder", "Bilinear",
    "Monotonic", "monotonic_prop", "get_lower_index", "Locator", "sharding", "device_count", "striped_ring", "output_empty", "output_zeros", "output_trim",
    "profile_enable", "profile_read", "PATH_AUTO", "PATH_GATHER", "PATH_BUCKETED",
]
