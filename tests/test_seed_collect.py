"""CPU tier: the oracle's restatement of the seed collection (oracle/seed_oracle.cpp: index image lookup, collect_seed_hits,
radix_sort_128x) against what the unmodified reference produced for the same minimizers and the same index image
(tests/golden/seeds/*.npz, made by tests/golden/make_seed_golden.py through oracle/_ref/mt_dump).  This pins the checker
the GPU tier compares the seed kernels with; no GPU involved, and nothing of the product runs here."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol

SEEDS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seeds", "*.npz")))


def load_fixture(path):
    g = np.load(path, allow_pickle=False)
    return {k: g[k] for k in g.files}


def test_fixtures_are_present():
    assert len(SEEDS) == 8


@pytest.mark.parametrize("path", SEEDS, ids=[os.path.basename(p)[:-4] for p in SEEDS])
def test_collect_seed_hits_matches_reference(path):
    g = load_fixture(path)
    n_reads = len(g["bid"])
    total = 0
    with ol.SeedIndex([g["img_B"], g["img_H"], g["img_V"], g["img_P"]]) as ix:
        for r in range(n_reads):
            mini = g["mini"][g["mini_off"][r]:g["mini_off"][r + 1]]
            a, rep_len, mini_pos = ix.collect_seeds(int(g["flag"]), int(g["mid_occ"]), g["bid"][r], g["qlen"][r], mini)
            exp = g["anchors"][g["a_off"][r]:g["a_off"][r + 1]]
            assert a.shape == exp.shape and np.array_equal(a, exp), (os.path.basename(path), r, "anchors (order of equal x included)")
            assert rep_len == int(g["rep_len"][r]), (r, "rep_len")
            assert np.array_equal(mini_pos, g["mini_pos"][g["mp_off"][r]:g["mp_off"][r + 1]]), (r, "mini_pos")
            total += len(a)
    assert total == len(g["anchors"])


def test_an_incomplete_image_is_refused():
    g = load_fixture(SEEDS[0])
    with pytest.raises(ValueError, match="incomplete"):
        ol.SeedIndex([g["img_B"], np.zeros(0, np.uint8), g["img_V"], g["img_P"]])
