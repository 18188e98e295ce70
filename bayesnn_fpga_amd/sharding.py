"""Multi-GPU: shard the T Monte-Carlo samples (or Masksembles mask indices) over ranks.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Samples are
independent given (seed, site, t) — the Philox counter carries t — so rank g runs
t in [lo_g, hi_g) on the same batch with the same seed, and the ONLY exchange is one all-reduce
(sum) of the float64 moment buffer [3, E, B, C] per batch (KBs: latency-bound, never link-bound;
SURVEY.md §8.5).  The deterministic prefix is recomputed per rank (cheaper than broadcasting it).  With fewer samples than
ranks the batch is partitioned by IMAGES instead (``partition`` / ``accumulate_partitioned``): no rank idles.
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


def partition(T, B, rank, world, kind=None):
    """What rank ``rank`` of ``world`` runs of a batch of B images x T Monte-Carlo samples: ``("samples", lo, hi)`` — the
    sample range [lo, hi) on all images — while there are more samples than ranks, else ``("images", lo, hi)`` —
    ALL T samples on images [lo, hi): the fallback of SURVEY.md §8.5 ("partition by images, pure DP") for T < G, where a
    sample split would leave ranks idle (T = 4 on 8 GPUs: half of them) and every rank would still recompute the whole
    once-per-batch prefix.  Either way the shares are disjoint and their moment buffers add up to the one-rank result.
    T == world (config 4: one Masksembles mask per GPU) also goes by images: measured on one MI355X, rank 0's share of eight as
    32 images x 8 samples takes 0.61 ms per batch against 0.77 ms as 250 images x 1 sample (tools/share_bench.py,
    profiles/r04_share_config4.txt: the prefix runs on 32 images instead of 250 and a head launch's 32-sample groups are full) —
    every GPU then applies all M masks to its images instead of one mask to all images; the reduce is the same one all-reduce.
    (T > world keeps the sample split the configs are written with — "T=512 sharded 64/GPU" — although the image split measured
    6 % faster there too: 3.00 vs 3.18 ms for T = 100 over eight ranks; ``kind`` / ``bench.py --partition`` force either.)"""
    if kind not in (None, "samples", "images"):
        raise ValueError(f"kind must be 'samples' or 'images', got {kind!r}")
    if kind is None:
        kind = "samples" if world == 1 or ((T > world or B < world) and T >= world) else "images"
    return (kind,) + (shard_range(T, rank, world) if kind == "samples" else shard_range(B, rank, world))


def _rank_world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def share_kind(engine, T, B, world, kind=None):
    """``partition``'s choice for this engine: with ``kind=None`` an image split that some rank's kernels cannot take (a share that
    does not start on a whole Philox call of every site, ``bmi_image_offset_ok``: e.g. B = 250 over 8 ranks with a 32-channel
    channel-wise site) falls back to the sample split — host-only and the same answer on every rank, so the choice stays
    collective-safe.  An explicit ``kind="images"`` that cannot be taken raises on every rank together."""
    chosen = partition(T, B, 0, world, kind)[0]
    if chosen != "images":
        return chosen
    # every rank checks EVERY rank's image offset: a partition is accepted or refused by the whole group before anyone launches,
    # instead of one rank raising while the others wait in the all-reduce
    bad = [r for r in range(world) if not engine.image_offset_ok(shard_range(B, r, world)[0])]
    if not bad:
        return "images"
    if kind is None:
        return "samples"
    raise ValueError(f"image partition of a batch of {B} over {world} ranks: the shares of ranks {bad} do not start on a "
                     "whole Philox call of every site (bmi_image_offset_ok)")


def accumulate_share(engine, x, S, T, seed=0, cnt0=0, rank=0, world=1, kind=None):
    """Rank ``rank``'s share of batch ``x`` x T samples ADDED into the moment buffer ``S`` [3, E, B, C]; NO collective (what a
    hipGraph of a rank's step captures: ``BatchesInFlight.predict_graphed``).  ``partition`` decides (``share_kind``): by samples
    while T > world size; by images when T <= world — the rank runs ``engine.accumulate(x[lo:hi], ..., image_offset=lo)``, masks
    drawn at the images' indices in the whole batch (bmi_forward_mcd_images), into its rows of S (the other ranks' rows stay zero
    until the all-reduce) — unless some rank's share cannot start where the split puts it: then by samples after all
    (``kind=None``), or a ValueError on every rank (explicit ``kind="images"``).  Nothing is allocated here: the share's
    [3, E, hi - lo, C] staging buffer belongs to the engine."""
    B = x.shape[0]
    kind = share_kind(engine, T, B, world, kind)
    _, lo, hi = partition(T, B, rank, world, kind)
    if kind == "samples":
        if hi > lo:
            engine.accumulate(x, S, lo, hi - lo, seed, cnt0)
        return S
    if hi > lo:
        cache = engine.__dict__.setdefault("_share_parts", {})
        key = (S.shape[1], hi - lo, S.shape[3])
        part = cache.get(key)
        if part is None:
            part = cache[key] = S.new_zeros(3, S.shape[1], hi - lo, S.shape[3])      # first call of this share shape only
        else:
            part.zero_()
        engine.accumulate(x[lo:hi], part, 0, T, seed, cnt0, image_offset=lo)
        S[:, :, lo:hi].add_(part)
    return S


def accumulate_partitioned(engine, x, S, T, seed=0, cnt0=0, group=None, kind=None, always_reduce=False):
    """This rank's share of batch ``x`` x T samples ADDED into the moment buffer ``S`` [3, E, B, C] (``accumulate_share``), then ONE
    all-reduce (sum) over the group.  ``always_reduce``: issue the collective in a group of ONE rank too (a sum over one rank: the same
    bits) — how a 1-GPU box exercises RCCL on exactly the streams and buffers the N-GPU path uses (bench.py's ``allreduce_us_1rank``)."""
    import torch.distributed as dist
    rank, world = _rank_world(group)
    accumulate_share(engine, x, S, T, seed, cnt0, rank, world, kind)
    if world > 1 or (always_reduce and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(S, op=dist.ReduceOp.SUM, group=group)
    return S


def predict_sharded(engine, x, T, seed=0, cnt0=0, group=None):
    """mean / var / logit_mean of T samples with the work partitioned over the process group (by samples, or by images
    when there are fewer samples than ranks)."""
    S = engine.new_moments(x.shape[0])
    accumulate_partitioned(engine, x, S, T, seed, cnt0, group)
    return engine.finalize(S, T)
