"""GPU tier: chains to hits on the GPU (csrc/chaindp_regs.hip) -- mm_gen_regs (hit.c:52-95) and
mm_est_err (esterr.c:30-64) over the chains chaindp_backtrack left in HBM, against the oracle's restatement (pinned to
the unmodified reference on the CPU tier, tests/test_regs.py).  mm_gen_regs is integer work and compared byte for byte;
of mm_est_err the counts n_match / n_tot and the div = -1 cases are exact, div itself (a logf) within 2e-6 relative."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from minimap2_chaindp_amd import anchorgen as ag, chaindp, params as P

pytestmark = pytest.mark.gpu
SEEDS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seeds", "*.npz")))
REGS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regs", "*.npz")))
DIV_RTOL = 2e-6


@pytest.fixture(scope="module")
def dev():
    with chaindp.Device(0, max_anchors=1 << 22, max_reads=1 << 12) as d:
        yield d


def check_div(got, n_match, n_tot, exp, e_match, e_tot, where):
    assert np.array_equal(n_match, e_match) and np.array_equal(n_tot, e_tot), (where, "n_match / n_tot")
    unset = exp["div"] < 0
    assert np.array_equal(got["div"] < 0, unset), (where, "hits without an estimate")
    assert np.array_equal(got["div"][unset], exp["div"][unset])
    assert np.allclose(got["div"][~unset], exp["div"][~unset], rtol=DIV_RTOL, atol=0), (where, "div")
    g2, e2 = got.copy(), exp.copy()
    g2["div"] = 0; e2["div"] = 0
    assert g2.tobytes() == e2.tobytes(), (where, "est_err touched something besides div")


@pytest.mark.parametrize("path", SEEDS, ids=[os.path.basename(p)[:-4] for p in SEEDS])
def test_fixture_minimizers_to_hits_without_leaving_the_device(dev, path):
    """The reference's own minimizers and index image in; seeds, DP, new_seed[], chains, hits and their divergence estimate
    on the GPU (mini_pos stays resident from the seed collection)."""
    g = np.load(path, allow_pickle=False)
    ix = dev.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    off, a, rep_len, mpo, mp = dev.collect_seeds(ix, int(g["flag"]), int(g["mid_occ"]), g["mini_off"], g["mini"], g["bid"], g["qlen"])
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    n_reads = len(g["qlen"])
    dev.run(par)
    dev.compact(par)
    coff, u, boff, b = dev.backtrack(par, pv[7])
    hash_ = (np.arange(n_reads, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(1 << 32)).astype(np.uint32)
    regs = dev.gen_regs(hash_, g["qlen"], coff[-1])
    exp = [ol.oracle_gen_regs(int(hash_[r]), int(g["qlen"][r]), u[coff[r]:coff[r + 1]], b[boff[r]:boff[r + 1]]) for r in range(n_reads)]
    for r in range(n_reads):
        assert regs[coff[r]:coff[r + 1]].tobytes() == exp[r].tobytes(), (r, "gen_regs")
    if not len(regs):
        return
    ref_len = (np.arange(int(regs["rid"].max()) + 1, dtype=np.int64) * 37 % 5000 + int(regs["re"].max()) - 2000).astype(np.int32)
    got, n_match, n_tot = dev.est_err(coff, regs, g["qlen"], ref_len)                      # resident mini_pos
    got2, n_match2, n_tot2 = dev.est_err(coff, regs, g["qlen"], ref_len, mpo, mp)           # the same, handed over by the host
    assert got.tobytes() == got2.tobytes() and np.array_equal(n_match, n_match2) and np.array_equal(n_tot, n_tot2)
    for r in range(n_reads):
        e, em, et = ol.oracle_est_err(ref_len, int(g["qlen"][r]), exp[r], b[boff[r]:boff[r + 1]], mp[mpo[r]:mpo[r + 1]])
        s = slice(int(coff[r]), int(coff[r + 1]))
        check_div(got[s], n_match[s], n_tot[s], e, em, et, r)


@pytest.mark.parametrize("path", REGS, ids=[os.path.basename(p)[:-4] for p in REGS])
def test_map_batch_minimizers_in_hits_out_equals_the_references_records(dev, path):
    """chaindp_map_batch: the reference's minimizers and index image in, its own mm_gen_regs records (tests/golden/regs/*.npz,
    made by the unmodified reference from the same reads) out, in one call with nothing else crossing PCIe -- and equal, byte
    for byte, to the stage-by-stage calls.  A second call with too little room reports the capacity and leaves the hits
    resident."""
    k = np.load(path, allow_pickle=False)
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(path)), "seeds", os.path.basename(path)), allow_pickle=False)
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    ix = dev.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    roff, regs, rep_len, n_anchors = dev.map_batch(ix, int(g["flag"]), int(g["mid_occ"]), par, int(k["min_cnt"]), g["mini_off"], g["mini"],
                                                  g["bid"], g["qlen"], k["hash"])
    assert np.array_equal(roff, k["chains_off"]) and n_anchors == len(g["anchors"]) and np.array_equal(rep_len, g["rep_len"])
    assert regs.tobytes() == k["regs"].view(ol.REG_DTYPE).reshape(-1).tobytes(), "mm_gen_regs records of the reference"
    if len(regs) > 1:
        roff2, regs2, _, _ = dev.map_batch(ix, int(g["flag"]), int(g["mid_occ"]), par, int(k["min_cnt"]), g["mini_off"], g["mini"],
                                           g["bid"], g["qlen"], k["hash"], regs_cap=1)             # too small: fetched afterwards
        assert np.array_equal(roff2, roff) and regs2.tobytes() == regs.tobytes()
    # the batch is resident as after the separate calls: the divergence estimate works on it, with the device's mini_pos
    got, n_match, n_tot = dev.est_err(roff, regs, k["qlen"], k["ref_len"])
    exp_div = k["regs_div"].view(ol.REG_DTYPE).reshape(-1)
    unset = exp_div["div"] < 0
    assert np.array_equal(got["div"] < 0, unset) and np.allclose(got["div"][~unset], exp_div["div"][~unset], rtol=DIV_RTOL, atol=0)


def test_map_batch_from_three_contexts_at_once(dev):
    """What bench.py's map_batch.three_contexts measures and the packet shim's service contexts do: chaindp_map_batch from three host
    threads, a context each, the index image shared -- every call returns the reference's records."""
    import threading
    path = REGS[-1]
    k = np.load(path, allow_pickle=False)
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(path)), "seeds", os.path.basename(path)), allow_pickle=False)
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    ix = dev.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    exp = k["regs"].view(ol.REG_DTYPE).reshape(-1).tobytes()
    devs = [chaindp.Device(0, max_anchors=len(g["anchors"]) + 1024, max_reads=len(g["bid"]) + 1) for _ in range(3)]
    bad = []

    def work(d):
        for _ in range(6):
            roff, regs, rep_len, n_anchors = d.map_batch(ix, int(g["flag"]), int(g["mid_occ"]), par, int(k["min_cnt"]), g["mini_off"], g["mini"],
                                                         g["bid"], g["qlen"], k["hash"])
            if not (np.array_equal(roff, k["chains_off"]) and regs.tobytes() == exp and np.array_equal(rep_len, g["rep_len"])):
                bad.append(1)
    try:
        th = [threading.Thread(target=work, args=(d,)) for d in devs]
        for t in th:
            t.start()
        for t in th:
            t.join()
    finally:
        for d in devs:
            d.close()
    assert not bad


@pytest.mark.parametrize("path", REGS, ids=[os.path.basename(p)[:-4] for p in REGS])
def test_hits_match_the_committed_reference_records(dev, path):
    """From the fixture's anchors to the reference's own mm_gen_regs / mm_est_err records (tests/golden/regs/*.npz, made by
    the unmodified reference): DP, new_seed[], chains and hits on the GPU."""
    k = np.load(path, allow_pickle=False)
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(path)), "seeds", os.path.basename(path)), allow_pickle=False)
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    dev.chain_batch(par, g["a_off"], g["anchors"])
    dev.compact(par)
    coff, u, boff, b = dev.backtrack(par, int(k["min_cnt"]))
    assert np.array_equal(coff, k["chains_off"]) and np.array_equal(u, k["u"]) and np.array_equal(b, k["b"].reshape(-1, 2))
    regs = dev.gen_regs(k["hash"], k["qlen"], coff[-1])
    exp, exp_div = k["regs"].view(ol.REG_DTYPE).reshape(-1), k["regs_div"].view(ol.REG_DTYPE).reshape(-1)
    assert regs.tobytes() == exp.tobytes(), "mm_gen_regs"
    got, n_match, n_tot = dev.est_err(coff, regs, k["qlen"], k["ref_len"], g["mp_off"], g["mini_pos"])
    unset = exp_div["div"] < 0
    assert np.array_equal(got["div"] < 0, unset) and np.allclose(got["div"][~unset], exp_div["div"][~unset], rtol=DIV_RTOL, atol=0)
    assert (n_match[~unset] >= 1).all() and (n_tot[~unset] >= n_match[~unset]).all()


def _mini_pos_for(rng, qlen, b):
    """A read's mini_pos consistent with its chained anchors: the forward query position of every anchor (so the search
    of esterr.c:44 finds them) and as many unrelated positions, sorted, with spans."""
    y = b[:, 1].astype(np.int64) & 0xffffffff
    span = (b[:, 1] >> np.uint64(32)).astype(np.int64) & 0xff
    rev = (b[:, 0] >> np.uint64(63)).astype(bool)
    pos = np.where(rev, qlen - 1 - (y + 1 - span), y)
    extra = rng.integers(0, max(qlen, 1), max(len(pos), 4))
    allp = np.unique(np.concatenate([pos, extra]))
    allp = allp[(allp >= 0) & (allp < (1 << 31))]
    return (rng.integers(10, 20, len(allp)).astype(np.uint64) << np.uint64(32)) | allp.astype(np.uint64)


@pytest.mark.parametrize("gen,preset,n_reads,min_cnt,over", [
    ("ava-ont", "ava-ont", 200, 3, {}),
    ("map-ont", "map-ont", 100, 3, {}),
    ("ties", "map-ont", 60, 1, dict(min_sc=0)),          # hundreds of chains per read: the radix procedure of the key sort
    ("paired", "sr", 800, 2, {}),
    ("skew", "ava-ont", 40, 3, {}),
])
def test_seeded_batches(dev, gen, preset, n_reads, min_cnt, over):
    par = P.preset(preset, **over)
    kw = dict(skew_max=40000) if gen == "skew" else {}
    off, a = ag.generate(gen, n_reads=n_reads, seed=77, **kw)
    dev.chain_batch(par, off, a)
    dev.compact(par)
    coff, u, boff, b = dev.backtrack(par, min_cnt)
    rng = np.random.default_rng(5)
    hash_ = rng.integers(0, 1 << 32, n_reads).astype(np.uint32)
    qlen = np.array([int((a[off[r]:off[r + 1], 1] & np.uint64(0xffffffff)).max()) + 50 if off[r + 1] > off[r] else 100 for r in range(n_reads)], np.int32)
    regs = dev.gen_regs(hash_, qlen, coff[-1])
    exp = [ol.oracle_gen_regs(int(hash_[r]), int(qlen[r]), u[coff[r]:coff[r + 1]], b[boff[r]:boff[r + 1]]) for r in range(n_reads)]
    for r in range(n_reads):
        assert regs[coff[r]:coff[r + 1]].tobytes() == exp[r].tobytes(), (gen, r)
    print(f"\n{gen}: {len(regs)} hits, most in one read {int(np.diff(coff).max())}")
    # est_err on a subset of the hits, as after chain_post (every other hit of a read dropped, the rest kept in order)
    keep = [np.arange(coff[r], coff[r + 1])[::2] for r in range(n_reads)]
    roff = np.concatenate([[0], np.cumsum([len(k) for k in keep])]).astype(np.int64)
    sub = regs[np.concatenate(keep)] if roff[-1] else regs[:0]
    mps = [_mini_pos_for(rng, int(qlen[r]), b[boff[r]:boff[r + 1]]) if r % 7 else np.zeros(0, np.uint64) for r in range(n_reads)]   # some reads without minimizers
    mpo = np.concatenate([[0], np.cumsum([len(m) for m in mps])]).astype(np.int64)
    mp = np.concatenate(mps) if mpo[-1] else np.zeros(0, np.uint64)
    ref_len = np.full(int(regs["rid"].max()) + 1 if len(regs) else 1, int(regs["re"].max()) + 30 if len(regs) else 1, np.int32)
    got, n_match, n_tot = dev.est_err(roff, sub, qlen, ref_len, mpo, mp)
    for r in range(n_reads):
        s = slice(int(roff[r]), int(roff[r + 1]))
        e, em, et = ol.oracle_est_err(ref_len, int(qlen[r]), sub[s], b[boff[r]:boff[r + 1]], mps[r])
        check_div(got[s], n_match[s], n_tot[s], e, em, et, (gen, r))


def test_regs_argument_errors(dev):
    par = P.preset("map-ont")
    off, a = ag.generate("map-ont", n_reads=4, seed=1)
    dev.chain_batch(par, off, a)
    with pytest.raises(chaindp.ChainDPError, match="backtrack"):        # chains of an earlier batch do not count
        dev.gen_regs(np.zeros(4, np.uint32), np.full(4, 100, np.int32), 0)
    dev.compact(par)
    coff, u, boff, b = dev.backtrack(par, 3)
    L, ctx = dev._lib, dev._ctx
    qlen = np.full(4, 1 << 20, np.int32)
    regs = dev.gen_regs(np.zeros(4, np.uint32), qlen, coff[-1])
    assert len(regs) == coff[-1] and len(regs) > 0
    assert L.chaindp_gen_regs(ctx, None, qlen.ctypes.data, regs.ctypes.data) != 0
    assert L.chaindp_gen_regs(ctx, np.zeros(4, np.uint32).ctypes.data, qlen.ctypes.data, None) != 0
    ref_len = np.full(int(regs["rid"].max()) + 1, 1 << 20, np.int32)
    with pytest.raises(chaindp.ChainDPError, match="mini_pos"):         # nothing resident: this batch was uploaded, not collected
        dev.est_err(coff, regs, qlen, ref_len)
    bad = regs.copy(); bad["as"][0] = 1 << 30
    with pytest.raises(chaindp.ChainDPError, match="beyond"):
        dev.est_err(coff, bad, qlen, ref_len, np.zeros(5, np.int64), np.zeros(0, np.uint64))
    bad_off = coff.copy(); bad_off[2] = bad_off[1] - 1 if bad_off[1] > 0 else -1
    with pytest.raises(chaindp.ChainDPError):
        dev.est_err(bad_off, regs, qlen, ref_len, np.zeros(5, np.int64), np.zeros(0, np.uint64))
    got, n_match, n_tot = dev.est_err(coff, regs, qlen, ref_len, np.zeros(5, np.int64), np.zeros(0, np.uint64))   # no minimizers: untouched
    assert got.tobytes() == regs.tobytes() and not n_match.any()
