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
