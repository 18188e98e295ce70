"""Multi-GPU: shard the T Monte-Carlo samples (or Masksembles mask indices) over ranks.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Samples are
independent given (seed, site, t) — the Philox counter carries t — so rank g runs
t in [lo_g, hi_g) on the same batch with the same seed, and the ONLY exchange is one all-reduce
(sum) of the float64 moment buffer [3, E, B, C] per batch (KBs: latency-bound, never link-bound;
SURVEY.md §8.5).  The deterministic prefix is recomputed per rank (cheaper than broadcasting it).
The reference has no counterpart: it is single-device (SA/train/train_utils.py:10-11).
"""
import torch


def shard_range(total, rank, world):
    """Contiguous, balanced split of range(total): the first ``total % world`` ranks get one extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(int(total), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def accumulate_sharded(accumulate_fn, S, T, group=None, t_begin=0):
    """Runs this rank's t-shard through ``accumulate_fn(S, t_lo, t_count)`` (which ADDS into the
    moment buffer ``S``) and sums the buffers over the group.  Works for any backend: the GPU path
    passes ``MCDEngine.accumulate`` and a RCCL group; the CPU tests pass a gloo group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(T, rank, world)
    if hi > lo:
        accumulate_fn(S, t_begin + lo, hi - lo)
    if world > 1:
        dist.all_reduce(S, op=dist.ReduceOp.SUM, group=group)
    return S


def predict_sharded(engine, x, T, seed=0, cnt0=0, group=None):
    """mean / var / logit_mean of T samples with the samples sharded over the process group."""
    S = engine.new_moments(x.shape[0])
    accumulate_sharded(lambda buf, t0, n: engine.accumulate(x, buf, t0, n, seed, cnt0), S, T, group)
    return engine.finalize(S, T)
