"""CPU tier: the N>1 path (read sharding + the two scalar reductions) with world_size 2 over gloo.
Each rank generates its shard of the seeded job, runs the ORACLE on it (no GPU here), and the ranks
reduce (max time, total anchors) exactly as bench.py does over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READS, SEED = 24, 77


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_skew(rank, world, port, q):
    """The skewed job (BASELINE configs[4]) dealt by anchor count: the code path `bench.py --gpus N --preset skew` takes."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from minimap2_chaindp_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, a = shard.generate_shard("skew", rank, world, 60, SEED, threads=1)
    dist.barrier()
    t_max, n_all = shard.reduce_job(0.5 + rank, int(off[-1]), dist)
    q.put((rank, len(off) - 1, int(off[-1]), int(a[:, 0].astype(np.int64).sum() & 0x7fffffff) if len(a) else 0, t_max, n_all))
    dist.destroy_process_group()


def test_two_ranks_deal_the_skewed_job_by_anchor_count():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_skew, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    from minimap2_chaindp_amd import anchorgen
    off, a = anchorgen.generate("skew", n_reads=120, seed=SEED, threads=2)
    (r0, n0, a0, c0, t0, s0), (r1, n1, a1, c1, t1, s1) = res
    assert n0 + n1 == 120 and a0 + a1 == int(off[-1]) == s0 == s1 and t0 == t1 == 1.5      # the shards tile the job
    cut = n0
    assert a0 == int(off[cut]) and c0 == int(a[:int(off[cut]), 0].astype(np.int64).sum() & 0x7fffffff)   # same reads, same anchors
    assert abs(a0 - a1) <= int(np.diff(off).max())                       # within one (largest) read of even, whatever the read counts


def _worker_strong(rank, world, port, q):
    """bench.py's default (--scaling strong): ONE fixed job cut over the ranks by anchor count (shard.generate_job_shard), the oracle
    as the compute leg, then the two reductions."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as ol
    from minimap2_chaindp_amd import params, shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, a, first = shard.generate_job_shard("ava-ont", rank, world, 2 * READS, SEED, threads=1)
    f, p, v, _ = ol.oracle_batch(params.preset("ava-ont"), off, a, threads=1)
    chk = int(np.bitwise_xor.reduce(f.astype(np.int64) * 31 + p))
    dist.barrier()
    t_max, n_all = shard.reduce_job(1.0 + rank, int(off[-1]), dist)
    q.put((rank, first, len(off) - 1, int(off[-1]), chk, t_max, n_all))
    dist.destroy_process_group()


def test_two_ranks_cut_one_fixed_job_by_anchor_count():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_strong, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from minimap2_chaindp_amd import anchorgen, params, shard
    off, a = anchorgen.generate("ava-ont", n_reads=2 * READS, seed=SEED, threads=2)
    f, p, v, _ = ol.oracle_batch(params.preset("ava-ont"), off, a, threads=2)
    cuts = shard.split_by_anchors(off, 2)
    assert [r[1] for r in res] == [int(cuts[0]), int(cuts[1])] and sum(r[2] for r in res) == 2 * READS      # the shards tile the job
    for rank, first, n_reads, n, chk, t_max, n_all in res:
        lo, hi = int(off[first]), int(off[first + n_reads])
        assert n == hi - lo and chk == int(np.bitwise_xor.reduce(f[lo:hi].astype(np.int64) * 31 + p[lo:hi]))   # same reads, same results
        assert t_max == 2.0 and n_all == int(off[-1])                                                     # max over ranks, sum over ranks
    assert abs(res[0][3] - res[1][3]) <= int(np.diff(off).max())                                          # near-equal anchor counts


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as ol
    from minimap2_chaindp_amd import params, shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, a = shard.generate_shard("ava-ont", rank, world, READS, SEED, threads=1)
    f, p, v, _ = ol.oracle_batch(params.preset("ava-ont"), off, a, threads=1)
    chk = int(np.bitwise_xor.reduce(f.astype(np.int64) * 31 + p))
    dist.barrier()
    t_max, n_all = shard.reduce_job(1.0 + rank, int(off[-1]), dist)
    q.put((rank, int(off[-1]), chk, t_max, n_all))
    dist.destroy_process_group()


def test_two_ranks_shard_and_reduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # single-process view of the same job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from minimap2_chaindp_amd import anchorgen, params
    off, a = anchorgen.generate("ava-ont", n_reads=2 * READS, seed=SEED, threads=2)
    f, p, v, _ = ol.oracle_batch(params.preset("ava-ont"), off, a, threads=2)
    for rank, n, chk, t_max, n_all in res:
        lo, hi = int(off[rank * READS]), int(off[(rank + 1) * READS])
        assert n == hi - lo                                             # shards tile the job
        assert chk == int(np.bitwise_xor.reduce(f[lo:hi].astype(np.int64) * 31 + p[lo:hi]))
        assert t_max == 2.0 and n_all == int(off[-1])                   # max over ranks, sum over ranks


def test_split_by_anchors_balances_skewed_batches():
    from minimap2_chaindp_amd import anchorgen, shard
    off, _ = anchorgen.generate("skew", n_reads=400, seed=5)
    cuts = shard.split_by_anchors(off, 8)
    assert cuts[0] == 0 and cuts[-1] == 400 and np.all(np.diff(cuts) >= 0)
    loads = np.diff(off[cuts])
    assert loads.max() < 1.35 * loads.mean() + np.diff(off).max()      # within one (largest) read of even
    with pytest.raises(ValueError):
        shard.shard_range(3, 2, 10)
