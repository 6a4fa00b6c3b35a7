"""CPU tier: the packet shim's host-side seed collection (csrc/seed_collect.cpp: index image lookup, collect_seed_hits,
radix_sort_128x) against what the unmodified reference produced for the same minimizers and the same index image
(tests/golden/seeds/*.npz, made by tests/golden/make_seed_golden.py through oracle/_ref/mt_dump).  No GPU involved."""
import glob
import os

import numpy as np
import pytest

from minimap2_chaindp_amd import chaindp, fpga

SEEDS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seeds", "*.npz")))


def load_fixture(path):
    g = np.load(path, allow_pickle=False)
    return {k: g[k] for k in g.files}


@pytest.fixture(autouse=True)
def fresh_index():
    yield
    fpga.lib().fpga_finalize()          # drops the index image (the driver is not up in this tier)


def test_fixtures_are_present():
    assert len(SEEDS) == 8


@pytest.mark.parametrize("path", SEEDS, ids=[os.path.basename(p)[:-4] for p in SEEDS])
def test_collect_seed_hits_matches_reference(path):
    g = load_fixture(path)
    fpga.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    fpga.lib().fpga_set_params(500, 0, 25, 40, int(g["flag"]), int(g["mid_occ"]))
    n_reads = len(g["bid"])
    total = 0
    for r in range(n_reads):
        mini = g["mini"][g["mini_off"][r]:g["mini_off"][r + 1]]
        a, rep_len, mini_pos = fpga.collect_seeds(g["bid"][r], g["qlen"][r], mini)
        exp = g["anchors"][g["a_off"][r]:g["a_off"][r + 1]]
        assert a.shape == exp.shape and np.array_equal(a, exp), (os.path.basename(path), r, "anchors (order of equal x included)")
        assert rep_len == int(g["rep_len"][r]), (r, "rep_len")
        assert np.array_equal(mini_pos, g["mini_pos"][g["mp_off"][r]:g["mp_off"][r + 1]]), (r, "mini_pos")
        total += len(a)
    assert total == len(g["anchors"])


def test_without_an_index_image_nothing_is_looked_up():
    with pytest.raises(chaindp.ChainDPError, match="index image"):
        fpga.collect_seeds(0, 100, np.zeros((3, 2), np.uint64))


def test_a_second_image_replaces_the_first():
    """main.c:201-204 sends B, H, V, P per index part; a B chunk after a complete image starts the next part's image."""
    g1, g2 = load_fixture(SEEDS[0]), load_fixture(SEEDS[-1])
    for g in (g1, g2, g1):
        fpga.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
        fpga.lib().fpga_set_params(500, 0, 25, 40, int(g["flag"]), int(g["mid_occ"]))
        a, _, _ = fpga.collect_seeds(g["bid"][0], g["qlen"][0], g["mini"][g["mini_off"][0]:g["mini_off"][1]])
        assert np.array_equal(a, g["anchors"][g["a_off"][0]:g["a_off"][1]])
