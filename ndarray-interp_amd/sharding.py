"""Query sharding over the GPUs of one node.

Two forms of the same split:
  * one process, several devices -- `interp_array_sharded` / `interp_array_ring_sharded`: replicas of one
    interpolator (one per device) behind ONE C-ABI call (ndi_interp{1,2}d_eval_sharded), one host thread per
    device inside the library.  This is what a Rust host does (benches/bench_interp1d.rs:49-79 is the shape);
  * one process per device -- `eval_sharded` over torch.distributed (what bench.py --gpus N uses).

Each query's result depends on read-only tables only (the reference's loop carries no state,
interp1d/mod.rs:334-342), so a batch shards embarrassingly: rank r of W evaluates the contiguous block
`shard_bounds(Q, r, W)` of the flattened query array on its own device, with knots / data / spline
tables replicated per device.  There is NO collective on the data path; the only cross-rank step is
reproducing the reference's *first-error* result: every rank reports its lowest failing global index
and the minimum over ranks wins (MIN all-reduce of one int64 -- RCCL on the GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from .errors import InterpolateError, Panic

NO_FAIL = (1 << 62)
DEVICE_FAILED = -1      # a rank whose evaluation died for a reason other than a failing query: wins every MIN


def shard_bounds(nq: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; block sizes differ by at most one."""
    base, rem = divmod(nq, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def eval_shard(evaluate, nq: int, rank: int, world: int):
    """Runs `evaluate(lo, hi)` for this rank's block and converts a local first-error into its global
    index.  Returns (global_fail_index or NO_FAIL, exception or None)."""
    lo, hi = shard_bounds(nq, rank, world)
    try:
        evaluate(lo, hi)
    except (InterpolateError.OutOfBounds, Panic) as e:  # local index -> global index
        # a panic (NaN query while extrapolating) ends the reference's loop at that query just like an Err
        return lo + (e.index if getattr(e, "index", None) is not None else 0), e
    except Exception as e:  # noqa: BLE001 -- device failure: this rank must still reach the all-reduce
        return DEVICE_FAILED, e   # (otherwise the other ranks block in it until the process-group timeout);
    return NO_FAIL, None          # it outranks every query index, so all ranks agree on the outcome


def eval_sharded(evaluate, nq: int, rank: int, world: int, group=None, device=None):
    """The whole protocol: evaluate this rank's block, MIN all-reduce the first failing global index (every
    rank always enters the collective), then the rank that owns the winning index re-raises its exception and
    the others raise `ShardFailed` naming it -- the reference's first-error result (interp1d/mod.rs:334-342)
    reproduced across ranks.  Returns normally when no rank failed."""
    local, exc = eval_shard(evaluate, nq, rank, world)
    first = first_error_across_ranks(local, group=group, device=device)
    if first == NO_FAIL:
        return
    if exc is not None and local == first:
        # every rank reports the same flat index of the whole batch (what the reference's serial loop would
        # name); the index within this rank's block stays available as .local_index
        if isinstance(exc, (InterpolateError.OutOfBounds, Panic)):
            exc.local_index = getattr(exc, "index", None)
            exc.index = first
        raise exc
    raise ShardFailed(first)


class ShardFailed(RuntimeError):
    """Another rank's shard holds the batch's first failing query."""

    def __init__(self, index: int):
        super().__init__(f"query {index} failed on another rank's shard" if index >= 0 else
                         "another rank's evaluation failed on its device")
        self.index = index


def first_error_across_ranks(local_fail: int, group=None, device=None) -> int:
    """MIN all-reduce of the per-rank first failing global index (NO_FAIL when a rank had none)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_fail
    t = torch.tensor([local_fail], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


# ------------------------------------------------------------------------------------------------------------
# one process, several devices: ndi_interp{1,2}d_eval_sharded / _eval_ring_sharded
# ------------------------------------------------------------------------------------------------------------
def _replica_handles(replicas):
    import ctypes as C
    from .interp2d import Interp2D
    strategies = [r.strategy for r in replicas]
    for st in strategies:
        if getattr(st, "_h", None) is None:
            raise TypeError("sharded evaluation needs built-in device strategies (f32 / f64 data)")
    two_d = isinstance(replicas[0], Interp2D)
    arr = (C.c_void_p * len(strategies))(*[st._h for st in strategies])
    return strategies, two_d, arr


def _flat_queries(strategies, xs, ys, two_d, io):
    """Queries of a sharded call: one host array (flattened; every shard reads its block of it) or a list with one
    device tensor per shard (that shard's block, resident on its device)."""
    from . import _capi
    from ._arrays import Buf
    dt = strategies[0]._np_dtype
    keep = []
    if isinstance(xs, (list, tuple)):
        if two_d and not (isinstance(ys, (list, tuple)) and len(ys) == len(xs)):
            raise TypeError("per-shard xs need per-shard ys")
        if len(xs) != len(strategies):
            raise TypeError("one query block per replica expected")
        nq = 0
        for i, t in enumerate(xs):
            b = Buf(t.reshape(-1), dt)
            keep.append(b)
            io[i].q = b.ptr
            nq += b.size
            if two_d:
                by = Buf(ys[i].reshape(-1), dt)
                keep.append(by)
                if by.size != b.size:
                    raise TypeError("`xs.shape()` and `ys.shape()` do not match")
                io[i].qy = by.ptr
        lo = 0
        for i, t in enumerate(xs):       # the blocks must be exactly the library's split of the whole batch
            a, b = shard_bounds(nq, i, len(xs))
            if b - a != keep[i * (2 if two_d else 1)].size:
                raise TypeError(f"shard {i}: block has {keep[i * (2 if two_d else 1)].size} queries, "
                                f"shard_bounds({nq}, {i}, {len(xs)}) is [{a}, {b})")
            lo = b
        return None, None, nq, keep[0].memspace, keep
    bx = Buf(xs.reshape(-1), dt)
    keep.append(bx)
    by = None
    if two_d:
        by = Buf(ys.reshape(-1), dt)
        keep.append(by)
        if tuple(xs.shape) != tuple(ys.shape):
            raise TypeError("`xs.shape()` and `ys.shape()` do not match")
    if bx.memspace != _capi.MEM_HOST:
        raise TypeError("a single flattened query array must be a host array; pass one device tensor per shard")
    return bx.ptr, (by.ptr if by is not None else None), bx.size, _capi.MEM_HOST, keep


def interp_array_sharded(replicas, xs, ys=None, *, out=None):
    """`interp_array` of one batch split over the replicas' devices in ONE library call.

    `replicas`: Interp1D (or Interp2D) objects built from the same arrays, one per device
    (`CubicSpline.new().device(d)` ...).  `xs` (`ys`): a host array of any rank, or one device tensor per shard
    holding `shard_bounds(nq, i, n)`'s block.  `out`: a host array of shape xs.shape ++ lanes (filled and
    returned), or None -- then every shard's rows stay on its device and the list of per-shard tensors of shape
    (rows_i, lanes) is returned.  First-error semantics are those of the reference's loop over the whole batch:
    the exception carries the global flat index, rows before it are written, later rows are untouched."""
    import ctypes as C
    import torch
    from . import _capi
    from .errors import raise_eval
    strategies, two_d, handles = _replica_handles(replicas)
    n = len(strategies)
    io = (_capi.ShardIO * n)()
    q, qy, nq, q_space, keep = _flat_queries(strategies, xs, ys, two_d, io)
    lanes, dt = strategies[0]._lanes, strategies[0]._np_dtype
    opts = _capi.EvalOpts()
    opts.q_memspace = q_space
    opts.path = strategies[0].path
    outs = None
    if out is not None:
        import numpy as np
        if not (isinstance(out, np.ndarray) and out.flags.c_contiguous and out.dtype == dt and out.size == nq * lanes):
            raise TypeError("out must be a C-contiguous host array of the data's element type and nq * lanes elements")
        opts.out_memspace = _capi.MEM_HOST
        for i in range(n):
            lo, _hi = shard_bounds(nq, i, n)
            io[i].out = out.ctypes.data + lo * lanes * out.itemsize
    else:
        tdt = torch.float64 if dt.itemsize == 8 else torch.float32
        opts.out_memspace = _capi.MEM_DEVICE
        outs = []
        for i, st in enumerate(strategies):
            lo, hi = shard_bounds(nq, i, n)
            t = torch.empty((hi - lo, lanes), dtype=tdt, device=f"cuda:{st._device}")
            outs.append(t)
            io[i].out = t.data_ptr() if hi > lo else None
            io[i].stream = torch.cuda.current_stream(st._device).cuda_stream
    info = _capi.OobInfo()
    lib = _capi.lib()
    if two_d:
        st = lib.ndi_interp2d_eval_sharded(handles, n, q, qy, nq, io, lanes, C.byref(opts), C.byref(info))
    else:
        st = lib.ndi_interp1d_eval_sharded(handles, n, q, nq, io, lanes, C.byref(opts), C.byref(info))
    del keep
    if st != _capi.OK:
        raise_eval(st, info)
    return out if out is not None else outs


def interp_array_ring_sharded(replicas, xs, ys=None, *, chunk_queries, consumer=None, n_slots=2, slots=None):
    """The ring evaluation (`Interp1D.interp_array_ring`) over several devices in one call: every shard streams
    its block of the batch through its own device-output ring.  `slots`: None (every handle owns a striped ring of
    `n_slots` slots) or one list of slot tensors per shard (see `striped_ring`).  `consumer(chunk, rows)` is called
    from the shards' host threads (serialised by the GIL here), once per chunk and in order within a shard;
    `chunk.shard` names the shard, `chunk.q_begin` is the global flat index; `rows` is the slot cut to the chunk's
    rows (None with library-owned rings)."""
    import ctypes as C
    import torch
    from . import _capi
    from .errors import raise_eval
    strategies, two_d, handles = _replica_handles(replicas)
    n = len(strategies)
    io = (_capi.ShardIO * n)()
    q, qy, nq, q_space, keep = _flat_queries(strategies, xs, ys, two_d, io)
    lanes = strategies[0]._lanes
    rings = (_capi.RingDesc * n)()
    keep_arr = []
    for i, st in enumerate(strategies):
        rings[i].chunk_queries = int(chunk_queries)
        io[i].stream = torch.cuda.current_stream(st._device).cuda_stream
        if slots is not None:
            arr = (C.c_void_p * len(slots[i]))(*[t.data_ptr() for t in slots[i]])
            keep_arr.append(arr)
            rings[i].slots = C.cast(arr, C.POINTER(C.c_void_p))
            rings[i].n_slots = len(slots[i])
            rings[i].row_stride = max(slots[i][0].stride(0), lanes)
        else:
            rings[i].n_slots = int(n_slots)
            rings[i].row_stride = lanes
    failed, keep_events = [], []

    def _cb(_user, cptr):
        if failed:
            return None
        try:
            c = cptr.contents
            view = slots[c.shard][c.slot][:c.q_count] if slots is not None else None
            ev = consumer(c, view)
        except BaseException as e:  # noqa: BLE001 -- must not unwind through the C frames
            failed.append(e)
            return None
        if ev is None:
            return None
        keep_events.append(ev)
        return ev.cuda_event
    cb = _capi.RING_CONSUMER(_cb) if consumer is not None else C.cast(None, _capi.RING_CONSUMER)
    opts = _capi.EvalOpts()
    opts.q_memspace = q_space
    opts.out_memspace = _capi.MEM_DEVICE
    opts.path = strategies[0].path
    info = _capi.OobInfo()
    lib = _capi.lib()
    if two_d:
        st = lib.ndi_interp2d_eval_ring_sharded(handles, n, q, qy, nq, io, rings, cb, None, C.byref(opts), C.byref(info))
    else:
        st = lib.ndi_interp1d_eval_ring_sharded(handles, n, q, nq, io, rings, cb, None, C.byref(opts), C.byref(info))
    del keep, keep_arr, keep_events
    if failed:
        raise failed[0]
    if st != _capi.OK:
        raise_eval(st, info)
