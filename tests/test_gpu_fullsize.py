"""GPU tier, BASELINE.json's full single-GPU sizes -- configs[3]'s per-GPU shard (the benchmark's own batch: 12,500 ava-ont
reads, ~76 M anchors), configs[2] (map-ont against a human-size reference, 24 targets, 32-bit positions, ~50 M anchors) and
configs[4] (skewed batch, 1e2 .. 1e5 anchors per read, the generator's full default range) -- checked through properties that
need no full-size oracle run:
  * local consistency: p[i] < i within the window, f[i] = f[p[i]] + pair score(i, p[i]) recomputed exactly
    (including the reference's f64 gap cost), f[i] = q_span where p[i] = -1, v[i] = max(f[i], v[p[i]]);
  * partition invariance: chaining two halves of the batch separately gives the same arrays (reads are
    independent) -- compared by checksum of checksums;
  * a random sample of reads against the oracle, bit for bit;
  * new_seed[]: offsets monotone, every record's predecessor index inside its read and smaller than its own.
"""
import numpy as np
import pytest

import oracle_lib as ol
from minimap2_chaindp_amd import chaindp, params as P, shard

pytestmark = pytest.mark.gpu

READS, SEED = 12_500, 20261004


def _checksum(*arrs):
    h = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        return _checksum_inner(h, arrs)


def _checksum_inner(h, arrs):
    for a in arrs:
        w = a.astype(np.int64).view(np.uint64)
        k = np.arange(1, len(w) + 1, dtype=np.uint64)
        h = (h * np.uint64(1099511628211)) ^ np.bitwise_xor.reduce(w * (k | np.uint64(1)))
    return int(h)


CONFIGS = [  # generator preset, DP preset, reads, least anchors expected
    ("ava-ont", "ava-ont", 12_500, 60_000_000),      # configs[3], one GPU's shard
    ("map-ont", "map-ont", 9_400, 45_000_000),       # configs[2]
    ("skew", "ava-ont", 3_000, 25_000_000),          # configs[4], skew_min = 100, skew_max = 100000 (the preset's defaults)
]


@pytest.mark.parametrize("gen,preset,READS,least", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_full_size_batch_properties(gen, preset, READS, least):
    par = P.preset(preset)
    off, a = shard.generate_shard(gen, 0, 1, READS, SEED, threads=16)
    tot = int(off[-1])
    assert tot > least
    if gen == "skew":
        n = np.diff(off)
        assert n.min() < 400 and n.max() > 50_000                    # the batch really spans the stated range
    with chaindp.Device(0, max_anchors=tot + 1, max_reads=READS + 1) as dev:
        f, p, v = dev.chain_batch(par, off, a)
        soff, seeds = dev.compact(par)
        # partition invariance (two half batches through the same context)
        mid = READS // 2
        f1, p1, v1 = dev.chain_batch(par, off[:mid + 1], a[:int(off[mid])])
        o2 = off[mid:] - off[mid]
        f2, p2, v2 = dev.chain_batch(par, o2, a[int(off[mid]):])
    assert _checksum(f, p, v) == _checksum(np.concatenate([f1, f2]), np.concatenate([p1, p2]), np.concatenate([v1, v2]))

    # ---- local consistency, vectorised over all anchors
    x, y = a[:, 0], a[:, 1]
    q = (y & np.uint64(0xffffffff)).astype(np.int64)
    span = ((y >> np.uint64(32)) & np.uint64(0xff)).astype(np.int64)
    n_per = np.diff(off)
    rs = np.repeat(off[:-1], n_per)                               # read start of every anchor
    idx = np.arange(tot, dtype=np.int64)
    has = p >= 0
    assert np.all(p[has] < (idx - rs)[has]) and np.all(p >= -1)
    assert np.array_equal(f[~has], span[~has].astype(np.int32))          # chain.c:251,283 with no predecessor
    j = rs[has] + p[has]
    i = idx[has]
    dr = (x[i] - x[j]).astype(np.int64)
    dq = q[i] - q[j]
    assert np.all((dr > 0) & (dr <= par.max_dist_x) & (dq > 0) & (dq <= min(par.max_dist_x, par.max_dist_y)))   # chain.c:252,257-258
    dd = np.abs(dr - dq)
    assert np.all(dd <= par.bw)                                          # chain.c:260
    sums = np.add.reduceat(span, off[:-1][n_per > 0])
    avg = np.zeros(READS, np.float32)
    avg[n_per > 0] = sums.astype(np.float32) / n_per[n_per > 0].astype(np.float32)      # chain.c:241 (f32 divide)
    avg_i = np.repeat(avg, n_per)[i].astype(np.float64)
    lin = (dd.astype(np.float64) * .01 * avg_i).astype(np.int64)          # chain.c:272, two f64 products, truncation
    lg = np.where(dd > 0, np.floor(np.log2(np.maximum(dd, 1))).astype(np.int64), 0)
    sc = np.minimum(np.minimum(dq, dr), span[i]) - (lin + (lg >> 1)) + f[j]
    assert np.array_equal(sc.astype(np.int32), f[i]), "f[i] != f[p[i]] + score(i, p[i])"
    vv = f.copy()
    vv[has] = np.maximum(f[has], v[j])
    assert np.array_equal(vv, v)                                          # chain.c:284

    # ---- a random sample of reads against the oracle
    rng = np.random.default_rng(1)
    for r in rng.choice(READS, 200, replace=False):
        lo, hi = int(off[r]), int(off[r + 1])
        of, op, ov, _ = ol.oracle_fpv(par, np.ascontiguousarray(a[lo:hi]))
        assert np.array_equal(of, f[lo:hi]) and np.array_equal(op, p[lo:hi]) and np.array_equal(ov, v[lo:hi]), int(r)
        exp = ol.oracle_compact(par, np.ascontiguousarray(a[lo:hi]), of, op, ov)
        assert seeds[int(soff[r]):int(soff[r + 1])].tobytes() == exp.tobytes(), int(r)

    # ---- new_seed[] structure
    assert soff[0] == 0 and np.all(np.diff(soff) >= 0) and int(soff[-1]) == len(seeds) <= tot
    assert np.all(np.diff(soff) <= n_per)
    pred = seeds["p"] >> 2
    own = np.arange(len(seeds), dtype=np.int64) - np.repeat(soff[:-1], np.diff(soff))
    assert np.all((pred < own) & (pred >= -1))
