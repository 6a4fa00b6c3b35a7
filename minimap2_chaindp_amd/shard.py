"""Read sharding across the GPUs of one node and the (only) collective steps of a multi-GPU run.

Reads are independent DP problems, so a job is partitioned, not exchanged.  Two ways to deal a seeded job:
  strong scaling (bench.py's default: BASELINE configs[3] is ONE job of 100,000 reads "read-sharded across 8"):
    generate_job_shard() -- the job is fixed, rank r of W takes the reads between the points where the cumulative
    ANCHOR count crosses r/W and (r+1)/W of the job's total, so per-GPU work shrinks as W grows;
  weak scaling: generate_shard() -- rank r owns reads [r*reads_per_gpu, (r+1)*reads_per_gpu), per-GPU work is fixed.
The only communication is the benchmark's barrier and two scalar reductions (max time, sum of anchors),
done with torch.distributed on whatever backend the process group uses (nccl = RCCL on the GPU box,
gloo in the CPU tests).
"""
import numpy as np

from . import anchorgen


def shard_range(rank, world, reads_per_gpu):
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return rank * reads_per_gpu, reads_per_gpu


def generate_shard(preset, rank, world, reads_per_gpu, seed, threads=None, balance=None, **overrides):
    """Rank `rank`'s share of the seeded job of world * reads_per_gpu reads.

    balance = "reads": reads [rank * reads_per_gpu, (rank + 1) * reads_per_gpu) -- right when reads are alike (ava-ont,
    map-ont).  balance = "anchors" (the default for the skewed preset, BASELINE configs[4]): the job is cut where the
    cumulative ANCHOR count crosses k/world of the total, so that a rank with a handful of 1e5-anchor reads does
    not hold up seven ranks of 1e2-anchor ones; every rank derives the same cuts from the job's offsets."""
    if balance is None:
        balance = "anchors" if preset == "skew" else "reads"
    if balance == "reads" or world == 1:
        first, n = shard_range(rank, world, reads_per_gpu)
    elif balance == "anchors":
        if not (0 <= rank < world):
            raise ValueError("rank outside world")
        cuts = split_by_anchors(anchorgen.offsets(preset, n_reads=world * reads_per_gpu, seed=seed, threads=threads, **overrides), world)
        first, n = int(cuts[rank]), int(cuts[rank + 1] - cuts[rank])
    else:
        raise ValueError("balance must be 'reads' or 'anchors'")
    return anchorgen.generate(preset, n_reads=n, seed=seed, first_read=first, threads=threads, **overrides)


def job_cuts(preset, world, job_reads, seed, threads=None, **overrides):
    """Read indices (len world+1) at which the seeded job of `job_reads` reads is cut for `world` ranks: every rank derives the
    same cuts from the job's offsets alone (no anchors generated, nothing exchanged)."""
    return split_by_anchors(anchorgen.offsets(preset, n_reads=job_reads, seed=seed, threads=threads, **overrides), world)


def generate_job_shard(preset, rank, world, job_reads, seed, threads=None, **overrides):
    """Strong scaling: rank `rank`'s share of the FIXED seeded job of `job_reads` reads, cut by anchor count.
    Returns (off, anchors, first_read): the shard's CSR offsets (starting at 0), its anchors, and the job-wide index of
    its first read."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    cuts = job_cuts(preset, world, job_reads, seed, threads=threads, **overrides)
    first, n = int(cuts[rank]), int(cuts[rank + 1] - cuts[rank])
    off, a = anchorgen.generate(preset, n_reads=n, seed=seed, first_read=first, threads=threads, **overrides)
    return off, a, first


def split_by_anchors(off, parts):
    """Cut points (read indices, len parts+1) that deal a batch's reads into `parts` contiguous pieces of
    near-equal anchor count (partition by cumulative anchors, not by read count)."""
    off = np.asarray(off, np.int64)
    total = int(off[-1])
    cuts = [0]
    for k in range(1, parts):
        cuts.append(int(np.searchsorted(off, total * k // parts, side="left")))
    cuts.append(len(off) - 1)
    return np.maximum.accumulate(np.array(cuts, np.int64))


def reduce_job(elapsed, anchors, dist=None, device=None):
    """(max over ranks of elapsed, sum over ranks of anchors).  dist = torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed), int(anchors)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([anchors], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())
