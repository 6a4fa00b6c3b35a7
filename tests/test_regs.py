"""CPU tier: the oracle's restatement of mm_gen_regs (hit.c:52-95) and mm_est_err (esterr.c:30-64) against the unmodified
reference (oracle/_ref/libmm2chain_ref.so, which compiles hit.c and esterr.c where they lie), on the chains of every seed
fixture (anchors and mini_pos as the reference's own front half produced them) and on seeded inputs with more than 64
chains per read (where the score sort leaves insertion sort for the radix procedure) and equal sort keys."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from minimap2_chaindp_amd import params as P

SEEDS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seeds", "*.npz")))
REGS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regs", "*.npz")))
needs_ref = pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")


def chains_of(par, a, min_cnt):
    f, p, v, _ = ol.oracle_fpv(par, a)
    seeds = ol.oracle_compact(par, a, f, p, v)
    return ol.oracle_bottom(min_cnt, par.min_sc, seeds)


def fixture_reads(path):
    g = np.load(path, allow_pickle=False)
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    for r in range(len(g["qlen"])):
        a = np.ascontiguousarray(g["anchors"][g["a_off"][r]:g["a_off"][r + 1]])
        mp = np.ascontiguousarray(g["mini_pos"][g["mp_off"][r]:g["mp_off"][r + 1]])
        yield par, pv[7], int(g["qlen"][r]), a, mp


@needs_ref
@pytest.mark.parametrize("path", SEEDS, ids=[os.path.basename(p)[:-4] for p in SEEDS])
def test_gen_regs_and_est_err_match_reference_on_fixtures(path):
    n_hits = n_div = 0
    for r, (par, min_cnt, qlen, a, mp) in enumerate(fixture_reads(path)):
        if not len(a):
            continue
        u, b = chains_of(par, a, min_cnt)
        for hash_ in (0, 0x9e3779b9 ^ r):
            regs = ol.oracle_gen_regs(hash_, qlen, u, b)
            exp = ol.ref_gen_regs(hash_, qlen, u, b)
            assert regs.tobytes() == exp.tobytes(), (r, hash_)
        n_hits += len(regs)
        if len(regs) and len(mp):
            ref_len = (np.arange(int(regs["rid"].max()) + 1, dtype=np.int64) * 37 % 5000 + regs["re"].max() - 2000).astype(np.int32)
            got, n_match, n_tot = ol.oracle_est_err(ref_len, qlen, regs, b, mp)
            exp = ol.ref_est_err(ref_len, qlen, regs, b, mp)
            assert got.tobytes() == exp.tobytes(), (r, "div")
            assert (n_match[got["div"] >= 0] >= 1).all()
            n_div += int((got["div"] >= 0).sum())
    if os.path.basename(path) != "mt_human_self_avaont.npz":      # a genome against itself under ava-ont's NO_DIAG: no chain survives
        assert n_hits > 0 and n_div > 0


@needs_ref
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gen_regs_with_many_chains_and_equal_keys(seed):
    """Hundreds of chains in one read (beyond the 64-element insertion sort of radix_sort_128x) and chains whose keys
    (score, count and scrambled low bits) coincide: identical copies of one chain on another target."""
    rng = np.random.default_rng(seed)
    n_u = int(rng.integers(70, 400))
    u, b = [], []
    for c in range(n_u):
        cnt = int(rng.integers(3, 12))
        rid, rev = int(rng.integers(0, 50)), int(rng.integers(0, 2))
        x0, y0 = int(rng.integers(100, 1 << 20)), int(rng.integers(100, 30000))
        xs = x0 + np.cumsum(rng.integers(1, 200, cnt))
        ys = y0 + np.cumsum(rng.integers(1, 200, cnt))
        span = rng.integers(5, 28, cnt)
        for i in range(cnt):
            b.append([(rev << 63) | (rid << 32) | int(xs[i]), (int(span[i]) << 32) | int(ys[i])])
        u.append((int(rng.integers(40, 60)) << 32) | cnt)
    u, b = np.array(u, np.uint64), np.array(b, np.uint64)
    # duplicate the first chain's first anchor into another chain's head: the hash part of the key is then equal too
    k1 = int(u[0] & np.uint64(0xffffffff))
    b[k1] = b[0]
    u[1] = (u[0] >> np.uint64(32) << np.uint64(32)) | (u[1] & np.uint64(0xffffffff))
    for hash_ in (0, 12345):
        regs = ol.oracle_gen_regs(hash_, 40000, u, b)
        exp = ol.ref_gen_regs(hash_, 40000, u, b)
        assert regs.tobytes() == exp.tobytes()
    assert (np.diff(regs["score"]) <= 0).all()


@pytest.mark.parametrize("path", REGS, ids=[os.path.basename(p)[:-4] for p in REGS])
def test_oracle_matches_the_committed_reference_hits(path):
    """tests/golden/regs/*.npz hold the unmodified reference's chains, mm_gen_regs records and mm_est_err results for the
    seed fixtures (tests/golden/make_regs_golden.py): the oracle reproduces them from the fixture's anchors alone, so the
    pin holds where oracle/_ref is not built."""
    k = np.load(path, allow_pickle=False)
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(path)), "seeds", os.path.basename(path)), allow_pickle=False)
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    exp, exp_div = k["regs"].view(ol.REG_DTYPE).reshape(-1), k["regs_div"].view(ol.REG_DTYPE).reshape(-1)
    for r in range(len(k["qlen"])):
        a = np.ascontiguousarray(g["anchors"][g["a_off"][r]:g["a_off"][r + 1]])
        u, b = chains_of(par, a, int(k["min_cnt"])) if len(a) else (np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64))
        c = slice(int(k["chains_off"][r]), int(k["chains_off"][r + 1]))
        assert np.array_equal(u, k["u"][c]) and np.array_equal(b.reshape(-1, 2), k["b"][k["b_off"][r]:k["b_off"][r + 1]]), (r, "chains")
        regs = ol.oracle_gen_regs(int(k["hash"][r]), int(k["qlen"][r]), u, b)
        assert regs.tobytes() == exp[c].tobytes(), (r, "mm_gen_regs")
        mp = np.ascontiguousarray(g["mini_pos"][g["mp_off"][r]:g["mp_off"][r + 1]])
        if len(regs) and len(mp):
            got, _, _ = ol.oracle_est_err(k["ref_len"], int(k["qlen"][r]), regs, b, mp)
            assert got.tobytes() == exp_div[c].tobytes(), (r, "mm_est_err")
