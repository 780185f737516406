"""Query sharding over the GPUs of one node.

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
        return lo, e        # (otherwise the other ranks block in it until the process-group timeout)
    return NO_FAIL, None


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
        raise exc
    raise ShardFailed(first)


class ShardFailed(RuntimeError):
    """Another rank's shard holds the batch's first failing query."""

    def __init__(self, index: int):
        super().__init__(f"query {index} failed on another rank's shard")
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
