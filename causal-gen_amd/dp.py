"""Data-parallel plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The HVAE shards naturally: samples are independent given the weights, so each rank trains on its own minibatch and
the only exchange per optimiser step is the average of the flat f32 gradient (8.2 / 31.9 / 69.5 MB for the
MNIST / mimic192 / ukbb192 models).  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is
per-link bound: a handful of large buckets beats 800 per-tensor calls by a wide margin, and each bucket's
collective is issued asynchronously so that RCCL overlaps them with one another and with the tail of the step.
The counterfactual loop needs no collective at all.
"""
import torch
import torch.distributed as dist


def bucketed_allreduce_mean(flat, bucket_elems, group=None, extra=()):
    """In-place average of `flat` (1-D) over the ranks of `group`, in buckets of `bucket_elems`; `extra` tensors
    (e.g. the [elbo, nll, kl] scalars, or a NaN flag) ride along.  Returns the number of collectives issued."""
    import os

    world = dist.get_world_size(group)
    if world == 1 and os.environ.get("CGEN_DP_FORCE") != "1":  # (CGEN_DP_FORCE=1: 1-rank dry run of the collectives themselves)
        return 0
    works = []
    for o in range(0, flat.numel(), bucket_elems):
        works.append(dist.all_reduce(flat[o:o + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for t in extra:
        works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    flat.mul_(1.0 / world)
    for t in extra:
        t.mul_(1.0 / world)
    return len(works)


def shared_categorical_draw(n=3):
    """One categorical per step for the whole GLOBAL batch (vae.py:310-319): drawn from the CPU generator, which
    every rank seeds identically (seed_all(args.seed)), so no collective is needed to agree on it."""
    return int(torch.distributions.Categorical(torch.ones(n) / n).sample())


def shard_batch(x, rank, world):
    """Equal contiguous shards (the reference uses drop_last=True, train_setup.py:35)."""
    per = x.shape[0] // world
    return x[rank * per:(rank + 1) * per]
