"""CPU tier: the from-scratch oracle (oracle/chain_oracle.c) and the wave-formulation model
(oracle/wave_model.c) against the golden vectors the unmodified reference produced
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import golden_names, load_golden, params_from


@pytest.mark.parametrize("name", golden_names())
def test_oracle_fpv_seeds_chains_match_reference_golden(name):
    g = load_golden(name)
    par = params_from(g["params"])
    off, a = g["off"], g["anchors"]
    for r in range(len(off) - 1):
        lo, hi = int(off[r]), int(off[r + 1])
        ar = np.ascontiguousarray(a[lo:hi])
        f, p, v, _ = ol.oracle_fpv(par, ar)
        assert np.array_equal(f, g["f"][lo:hi]), (name, r, "f")
        assert np.array_equal(p, g["p"][lo:hi]), (name, r, "p")
        assert np.array_equal(v, g["v"][lo:hi]), (name, r, "v")
        seeds = ol.oracle_compact(par, ar, f, p, v)
        slo, shi = int(g["seeds_off"][r]), int(g["seeds_off"][r + 1])
        assert seeds.tobytes() == g["seeds"][slo * 24:shi * 24].tobytes(), (name, r, "new_seed[]")
        u, b = ol.oracle_bottom(int(g["min_cnt"]), par.min_sc, seeds)
        ulo, uhi = int(g["chains_u_off"][r]), int(g["chains_u_off"][r + 1])
        blo, bhi = int(g["chains_b_off"][r]), int(g["chains_b_off"][r + 1])
        assert np.array_equal(u, g["chains_u"][ulo:uhi]), (name, r, "u")
        assert np.array_equal(b, g["chains_b"][blo:bhi]), (name, r, "b")


@pytest.mark.parametrize("ring", [64, 128, 256])
@pytest.mark.parametrize("name", golden_names())
def test_wave_model_matches_golden(name, ring):
    g = load_golden(name)
    par = params_from(g["params"])
    f, p, v, stats = ol.wave_model_batch(par, g["off"], np.ascontiguousarray(g["anchors"]), ring=ring)
    assert np.array_equal(f, g["f"]) and np.array_equal(p, g["p"]) and np.array_equal(v, g["v"]), (name, stats)


def test_mt_fixture_is_the_survey_case():
    """BASELINE config 1: MT-orang vs MT-human, map-ont -> 346 anchors, 1 chain, score 3189, 342 anchors."""
    g = load_golden("mt_orang_human_mapont")
    assert g["anchors"].shape == (346, 2)
    assert tuple(g["params"]) == (5000, 5000, 500, 25, 40, 0, 1)
    assert len(g["chains_u"]) == 1
    assert int(g["chains_u"][0]) >> 32 == 3189 and int(g["chains_u"][0]) & 0xffffffff == 342
    _, _, _, evals = ol.oracle_fpv(params_from(g["params"]), np.ascontiguousarray(g["anchors"]))
    assert evals == 8970                      # inner-loop executions measured on the reference


def test_toy_colinear_known_answer():
    """six colinear anchors -> f = 15,30,...,90, p = -1,0,...,4, new_seed.p = -4|flags..."""
    g = load_golden("edge_cases_mapont")
    lo, hi = int(g["off"][2]), int(g["off"][3])
    assert list(g["f"][lo:hi]) == [15, 30, 45, 60, 75, 90]
    assert list(g["p"][lo:hi]) == [-1, 0, 1, 2, 3, 4]


def test_empty_and_single():
    par = params_from([5000, 5000, 500, 25, 40, 0, 1])
    f, p, v, ev = ol.oracle_fpv(par, np.zeros((0, 2), np.uint64))
    assert len(f) == 0 and ev == 0
    a = np.array([[10, (15 << 32) | 20]], np.uint64)
    f, p, v, ev = ol.oracle_fpv(par, a)
    assert (f[0], p[0], v[0], ev) == (15, -1, 15, 0)
    assert len(ol.oracle_compact(par, a, f, p, v)) == 0           # v < min_sc and no predecessor: dropped
    par.min_sc = 15
    s = ol.oracle_compact(par, a, f, p, v)
    assert len(s) == 1 and s["p"][0] == -3 and s["f"][0] == 15    # (-1<<2)|1


def test_batch_threads_agree():
    from minimap2_chaindp_amd import anchorgen as ag, params as P
    off, a = ag.generate("ava-ont", n_reads=12, seed=3)
    par = P.preset("ava-ont")
    f1, p1, v1, e1 = ol.oracle_batch(par, off, a, threads=1)
    f4, p4, v4, e4 = ol.oracle_batch(par, off, a, threads=5)
    assert e1 == e4 and np.array_equal(f1, f4) and np.array_equal(p1, p4) and np.array_equal(v1, v4)
    for r in (0, 5, 11):
        fr, pr, vr, _ = ol.oracle_fpv(par, np.ascontiguousarray(a[off[r]:off[r + 1]]))
        assert np.array_equal(fr, f1[off[r]:off[r + 1]]) and np.array_equal(pr, p1[off[r]:off[r + 1]])
