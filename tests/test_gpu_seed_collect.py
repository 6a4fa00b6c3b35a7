"""GPU tier: seed collection on the GPU (csrc/chaindp_seed.hip: index-image lookup, skip_seed, expansion, rep_len, mini_pos,
the reference's radix_sort_128x per read) against what the unmodified reference produced for the same minimizers and index
image (tests/golden/seeds/*.npz), and then straight into the chaining DP without the anchors leaving HBM."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from minimap2_chaindp_amd import chaindp, params as P

pytestmark = pytest.mark.gpu
SEEDS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seeds", "*.npz")))


@pytest.fixture(scope="module")
def dev():
    with chaindp.Device(0, max_anchors=1 << 20, max_reads=1 << 10) as d:
        yield d


@pytest.mark.parametrize("path", SEEDS, ids=[os.path.basename(p)[:-4] for p in SEEDS])
def test_gpu_collect_seed_hits_matches_reference_and_feeds_the_dp(dev, path):
    g = np.load(path, allow_pickle=False)
    ix = dev.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    off, a, rep_len, mpo, mp = dev.collect_seeds(ix, int(g["flag"]), int(g["mid_occ"]), g["mini_off"], g["mini"], g["bid"], g["qlen"])
    assert np.array_equal(off, g["a_off"]), "anchors per read"
    assert np.array_equal(a, g["anchors"]), "anchors, order of equal x included"
    assert np.array_equal(rep_len, g["rep_len"]) and np.array_equal(mpo, g["mp_off"]) and np.array_equal(mp, g["mini_pos"])
    # the resident anchors chain like uploaded ones
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    dev.run(par)
    f, p, v = dev.download()
    if len(a):
        of, op, ov, _ = ol.oracle_batch(par, off, np.ascontiguousarray(g["anchors"]), threads=4)
        assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov)


def test_gpu_seed_collection_edge_cases(dev):
    g = np.load(SEEDS[0], allow_pickle=False)
    ix = dev.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    # no reads; reads without minimizers; a read whose minimizers are all absent from the index
    off, a, rep, mpo, mp = dev.collect_seeds(ix, 0, 50, np.zeros(1, np.int64), np.zeros((0, 2), np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.int32))
    assert list(off) == [0] and len(a) == 0
    absent = np.array([[(0x123456789 << 8) | 15, 40], [(0x1234567ab << 8) | 15, 90]], np.uint64)
    off, a, rep, mpo, mp = dev.collect_seeds(ix, 0, 50, np.array([0, 0, 2, 2], np.int64), absent, np.zeros(3, np.uint32), np.full(3, 1000, np.int32))
    assert list(off) == [0, 0, 0, 0] and list(mpo) == [0, 0, 2, 2] and list(rep) == [0, 0, 0]
    assert list(mp) == [(15 << 32) | 20, (15 << 32) | 45]
    # capacity: more seeds than the context can hold is an error, not a truncation
    big = np.load(SEEDS[-1], allow_pickle=False)
    with chaindp.Device(0, max_anchors=64, max_reads=64) as small:
        ixs = small.load_index([big["img_B"], big["img_H"], big["img_V"], big["img_P"]])
        with pytest.raises(chaindp.ChainDPError, match="capacity"):
            small.collect_seeds(ixs, int(big["flag"]), int(big["mid_occ"]), big["mini_off"], big["mini"], big["bid"], big["qlen"])


BIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "_big", "big_avaont.npz")


@pytest.mark.skipif(not os.path.exists(BIG), reason="large reference dump not present (git-ignored; made in the build container)")
def test_gpu_seed_collection_at_scale():
    """600 reads x 8 kb all-vs-all through the reference's own indexer and collect_seed_hits: 1.6 M minimizers, 2.9 M anchors."""
    import time
    g = np.load(BIG, allow_pickle=False)
    with chaindp.Device(0, max_anchors=1 << 23, max_reads=1 << 12) as d:
        ix = d.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
        for rep in range(3):
            t0 = time.time()
            off, a, rep_len, mpo, mp = d.collect_seeds(ix, int(g["flag"]), int(g["mid_occ"]), g["mini_off"], g["mini"], g["bid"], g["qlen"])
            dt = time.time() - t0
        print(f"\nseed collection at scale: {len(a)} anchors from {len(g['mini'])} minimizers in {dt * 1e3:.1f} ms (incl. transfers) "
              f"-> {len(a) / dt / 1e6:.1f} M anchors/s")
        assert np.array_equal(off, g["a_off"]) and np.array_equal(a, g["anchors"])
        assert np.array_equal(rep_len, g["rep_len"]) and np.array_equal(mp, g["mini_pos"])
