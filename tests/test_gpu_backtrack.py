"""GPU tier: mm_chain_dp_bottom (reference chain.c:329-431) on the GPU against the chains the
unmodified reference produced (golden fixtures) and against the oracle's restatement on seeded batches:
u[] (score<<32|count) and the chained anchors b[], per read, element for element."""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import golden_names, load_golden, params_from
from minimap2_chaindp_amd import anchorgen as ag, chaindp, params as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    with chaindp.Device(0, max_anchors=1 << 22, max_reads=1 << 15) as d:
        yield d


@pytest.mark.parametrize("name", golden_names())
def test_golden_chains(dev, name):
    g = load_golden(name)
    par = params_from(g["params"])
    dev.chain_batch(par, g["off"], g["anchors"])
    dev.compact(par)
    coff, u, boff, b = dev.backtrack(par, int(g["min_cnt"]))
    assert np.array_equal(coff, g["chains_u_off"]), (name, "chains per read")
    assert np.array_equal(u, g["chains_u"]), (name, "u")
    assert np.array_equal(boff, g["chains_b_off"]), (name, "anchors per read")
    assert np.array_equal(b, g["chains_b"].reshape(-1, 2)), (name, "b")


@pytest.mark.parametrize("gen,preset,n_reads,min_cnt,over", [
    ("ava-ont", "ava-ont", 200, 3, {}),
    ("map-ont", "map-ont", 100, 3, {}),
    ("ties", "map-ont", 100, 3, {}),
    ("ties", "map-ont", 60, 1, dict(min_sc=0)),          # every anchor chains: many chains, shared peaks
    ("paired", "sr", 1500, 2, {}),
    ("dense", "ava-ont", 2, 3, dict()),
    ("skew", "ava-ont", 60, 3, {}),
])
def test_chains_match_oracle(dev, gen, preset, n_reads, min_cnt, over):
    par = P.preset(preset, **over)
    kw = dict(read_len=2500, n_hits=10) if gen == "dense" else (dict(skew_max=60000) if gen == "skew" else {})
    off, a = ag.generate(gen, n_reads=n_reads, seed=2024, **kw)
    f, p, v = dev.chain_batch(par, off, a)
    soff, seeds = dev.compact(par)
    coff, u, boff, b = dev.backtrack(par, min_cnt)
    for r in range(n_reads):
        eu, eb = ol.oracle_bottom(min_cnt, par.min_sc, seeds[int(soff[r]):int(soff[r + 1])])
        assert np.array_equal(u[int(coff[r]):int(coff[r + 1])], eu), (gen, r, "u")
        assert np.array_equal(b[int(boff[r]):int(boff[r + 1])], eb.reshape(-1, 2)), (gen, r, "b")


def test_many_chains_with_equal_first_x(dev):
    """More than 64 chains in one read, several starting at the same x: the order among them is whatever the
    reference's unstable radix sort leaves (ksort.h:101-151), which the GPU path reproduces."""
    rng = np.random.default_rng(3)
    rows = []
    for c in range(150):                                     # 150 short colinear runs, far apart in q, x starts repeated
        x0 = 1000 + 40 * (c % 30)                            # only 30 distinct starting x
        q0 = 100_000 * c                                     # far apart on the query: runs never chain with each other
        for k in range(4):
            rows.append(((3 << 32) | (x0 + 9 * k), (15 << 32) | (q0 + 9 * k)))
    a = np.array(sorted(rows), np.uint64)
    off = np.array([0, len(a)], np.int64)
    par = P.preset("map-ont", min_sc=20, max_dist_y=50, max_dist_x=5000)
    f, p, v = dev.chain_batch(par, off, a)
    soff, seeds = dev.compact(par)
    coff, u, boff, b = dev.backtrack(par, 2)
    eu, eb = ol.oracle_bottom(2, par.min_sc, seeds)
    assert len(eu) > 64
    assert np.array_equal(u, eu) and np.array_equal(b, eb.reshape(-1, 2))
    if ol.have_ref():
        ru, rb = ol.ref_bottom(2, par.min_sc, 1, seeds)
        assert np.array_equal(u, ru) and np.array_equal(b, rb.reshape(-1, 2))
