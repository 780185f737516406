"""Array plumbing between numpy / torch and the C ABI (pointers + memory space)."""
from __future__ import annotations

import numpy as np

from . import _capi

try:  # torch is plumbing for device memory and streams only
    import torch
except Exception:  # pragma: no cover
    torch = None


def is_torch(a) -> bool:
    return torch is not None and isinstance(a, torch.Tensor)


_NP2ID = {np.dtype(np.float32): _capi.F32, np.dtype(np.float64): _capi.F64}


def np_dtype_of(a):
    if is_torch(a):
        return {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64)}.get(a.dtype) \
            or np.dtype(str(a.dtype).replace("torch.", ""))
    return np.asarray(a).dtype


def dtype_id(dt) -> int:
    dt = np.dtype(dt)
    if dt not in _NP2ID:
        raise TypeError(f"the MI355X path covers float32/float64 only, got {dt} "
                        "(other element types stay on the host's generic per-query path)")
    return _NP2ID[dt]


class Buf:
    """A contiguous buffer handed to the C ABI: pointer, memory space and a keep-alive."""

    def __init__(self, arr, dt=None):
        if is_torch(arr):
            t = arr if dt is None else arr.to({np.dtype(np.float32): torch.float32,
                                               np.dtype(np.float64): torch.float64}[np.dtype(dt)])
            t = t.contiguous()
            self.keep = t
            self.shape = tuple(t.shape)
            self.size = t.numel()
            if t.is_cuda:
                self.memspace = _capi.MEM_DEVICE
                self.device = t.device.index if t.device.index is not None else torch.cuda.current_device()
                self.ptr = t.data_ptr()
            else:
                self.memspace = _capi.MEM_HOST
                self.device = None
                self.ptr = t.data_ptr()
            self.np_dtype = np_dtype_of(t)
        else:
            a = np.ascontiguousarray(arr, dtype=dt)
            self.keep = a
            self.shape = a.shape
            self.size = a.size
            self.memspace = _capi.MEM_HOST
            self.device = None
            self.ptr = a.ctypes.data
            self.np_dtype = a.dtype


def current_stream_ptr(device: int):
    if torch is None or not torch.cuda.is_available():
        return None
    return torch.cuda.current_stream(device).cuda_stream



def striped_ring(chunk_queries: int, lanes: int, n_slots: int, dtype=np.float64, device: int = 0):
    """The recommended ring layout on MI355X (include/ndinterp.h): ONE device allocation with the
    slots interleaved row by row.  Returns `n_slots` views of shape (chunk_queries, lanes) whose rows are
    contiguous and `n_slots * lanes` elements apart -- every chunk's output stream then covers the whole ring's
    physical extent instead of one 1/n_slots part of it."""
    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(dtype)]
    base = torch.empty((chunk_queries, n_slots, lanes), dtype=tdt, device=f"cuda:{device}")
    return [base[:, s, :] for s in range(n_slots)]


class _OwnedOutput:
    """A buffer from ndi_output_alloc exposed through __cuda_array_interface__: torch.as_tensor() wraps it without a copy
    and keeps this object alive; the buffer goes back with ndi_output_free when the last tensor over it is gone."""

    def __init__(self, shape, dtype, device, max_tries=0, zeroed=True):
        import ctypes
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        ptr = ctypes.c_void_p()
        self.info = _capi.OutputInfo()
        st = _capi.lib().ndi_output_alloc(self.device, nbytes, int(max_tries),
                                          _capi.OUTPUT_ZEROED if zeroed else _capi.OUTPUT_UNINITIALIZED,
                                          ctypes.byref(ptr), ctypes.byref(self.info))
        if st != _capi.OK:
            from .errors import DeviceError
            raise DeviceError(_capi.last_error())
        self.ptr = ptr.value
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False),
                                         "version": 2, "strides": None}

    def __del__(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p:
            try:
                _capi.lib().ndi_output_free(p)
            except Exception:   # interpreter shutdown
                pass


OUTPUT_OWNED_MIN_BYTES = 1 << 30


def output_trim() -> None:
    """Release the output buffers ndi_output_free keeps for reuse (ndi_output_trim)."""
    _capi.lib().ndi_output_trim()


def output_empty(shape, dtype=np.float64, device: int = 0, max_tries: int = 0, zeroed: bool = False):
    """A library-owned device output buffer (ndi_output_alloc: the allocation of interp_array, interp1d/mod.rs:209, with the
    placement check of include/ndinterp.h) as a torch tensor of `shape`.  `tensor.ndi_output_info` tells how many candidates
    were tried and the fill rate of the one kept.  Contents unspecified (NDI_OUTPUT_UNINITIALIZED: a buffer kept by
    ndi_output_free comes back without a refill -- interp_array overwrites every row or drops the buffer); `output_zeros`
    is the reference's Array::zeros."""
    own = _OwnedOutput(shape, dtype, device, max_tries, zeroed)
    with torch.cuda.device(device):
        t = torch.as_tensor(own, device=f"cuda:{device}")
    t.ndi_output_info = {"tries": own.info.tries, "fill_TBps": round(own.info.fill_tbps, 3),
                         "worst_fill_TBps": round(own.info.worst_fill_tbps, 3), "alloc_ms": round(own.info.alloc_ms, 2)}
    return t


def output_zeros(shape, dtype=np.float64, device: int = 0, max_tries: int = 0):
    """`output_empty` filled with zeros: Array::zeros (interp1d/mod.rs:209)."""
    return output_empty(shape, dtype, device, max_tries, zeroed=True)
