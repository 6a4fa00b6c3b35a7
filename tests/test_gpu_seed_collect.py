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
        # ... and on to new_seed[] and the chains of mm_chain_dp_bottom, all without leaving the device
        soff, seeds = dev.compact(par)
        coff, u, boff, b = dev.backtrack(par, pv[7])
        for r in range(len(off) - 1):
            ar = np.ascontiguousarray(g["anchors"][off[r]:off[r + 1]])
            exp = ol.oracle_compact(par, ar, of[off[r]:off[r + 1]].copy(), op[off[r]:off[r + 1]].copy(), ov[off[r]:off[r + 1]].copy())
            assert seeds[int(soff[r]):int(soff[r + 1])].tobytes() == exp.tobytes(), (r, "new_seed")
            eu, eb = ol.oracle_bottom(pv[7], par.min_sc, exp)
            assert np.array_equal(u[int(coff[r]):int(coff[r + 1])], eu) and np.array_equal(b[int(boff[r]):int(boff[r + 1])], eb.reshape(-1, 2)), (r, "chains")


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


def _build_image(rng, n_keys, max_cnt, b_bits=6, rid_pool=None, pos_bits=21):
    """A synthetic index image in the reference's FPGA layout (index.c:603-720): random minimizers with 1..max_cnt positions,
    hashed into 2^b_bits buckets with khash's own probing (khash.h:218-231).  Returns (blobs, minimizer values)."""
    keys = rng.choice(1 << 34, size=n_keys, replace=False).astype(np.uint64) + np.uint64(1)
    buckets = [[] for _ in range(1 << b_bits)]
    for m in keys:
        buckets[int(m) & ((1 << b_bits) - 1)].append(int(m))
    B, H, V, Pa = bytearray(), bytearray(), bytearray(), bytearray()
    allh = allp = 0
    for bk in buckets:
        if not bk:
            B += (0).to_bytes(16, "little")
            continue
        nb = 4
        while nb < 2 * len(bk):
            nb <<= 1
        slots_k, slots_v, used, p_local = [0] * nb, [0] * nb, [False] * nb, []
        for m in bk:
            cnt = int(rng.integers(1, max_cnt + 1))
            pos = [int(rng.integers(0, rid_pool or 1 << 20)) << 43 | int(rng.integers(0, 1 << pos_bits)) << 22 | int(rng.integers(0, 2)) << 21 | int(rng.integers(0, 1 << 10))
                   for _ in range(cnt)]
            key = (m >> b_bits) << 1
            i, step = (key >> 1) & (nb - 1), 0
            while used[i]:
                step += 1
                i = (i + step) & (nb - 1)
            used[i] = True
            if cnt == 1:
                slots_k[i], slots_v[i] = key | 1, pos[0]
            else:
                slots_k[i], slots_v[i] = key, len(p_local) << 32 | cnt
                p_local += pos
        tmp_nb = (nb + 7) & ~7
        B += (((allp & 0xff) << 56) | (nb << 24)).to_bytes(8, "little") + ((allh << 28) | (allp >> 8)).to_bytes(8, "little")
        flags = [0] * max(1, nb >> 4)
        for i in range(nb):
            if not used[i]:
                flags[i >> 4] |= 2 << ((i & 15) << 1)                 # "empty" (khash.h:166)
        for g0 in range(0, tmp_nb, 8):
            H += (flags[g0 >> 4] & 0xffffffff).to_bytes(4, "little")
            for i in range(g0, g0 + 8):
                H += ((slots_k[i] if i < nb else 0) & 0xffffffffffff).to_bytes(6, "little")
                V += ((slots_v[i] if i < nb else 0)).to_bytes(8, "little")
            H += bytes(12)
        for x in p_local:
            Pa += x.to_bytes(8, "little")
        allh += tmp_nb
        allp += len(p_local)
    blobs = [np.frombuffer(bytes(x), np.uint8).copy() for x in (B, H, V, Pa)]
    return blobs, keys


def _synthetic_case(seed, max_cnt, n_mini, rep_pct, rid_pool=None, pos_bits=21, quiet=False):
    rng = np.random.default_rng(seed)
    blobs, keys = _build_image(rng, 3000, max_cnt, rid_pool=rid_pool, pos_bits=pos_bits)
    n_reads, flag, max_occ = 6, int(rng.choice([0, 3, 0x100000])), int(max_cnt * 3 // 4 + 2)
    mini, mini_off, bid, qlen = [], [0], [], []
    for r in range(n_reads):
        nm = int(n_mini * (r + 1) / n_reads)
        pool = keys[rng.integers(0, len(keys), max(2, nm * (100 - rep_pct) // 100))]
        pick = pool[rng.integers(0, len(pool), nm)]
        absent = rng.random(nm) < 0.05
        pick = np.where(absent, pick + np.uint64(1 << 36), pick)
        span = rng.integers(1, 29, nm).astype(np.uint64)
        qp = np.sort(rng.integers(0, 30000, nm)).astype(np.uint64)
        x = pick << np.uint64(8) | span
        y = (qp << np.uint64(1)) | rng.integers(0, 2, nm).astype(np.uint64)
        mini.append(np.stack([x, y], 1)); mini_off.append(mini_off[-1] + nm)
        bid.append(int(rng.integers(0, 1 << 10)) | (int(rng.integers(0, 2)) << 31)); qlen.append(30100)
    mini = np.concatenate(mini)
    with ol.SeedIndex(blobs) as oix:
        exp = [oix.collect_seeds(flag, max_occ, bid[r], qlen[r], mini[mini_off[r]:mini_off[r + 1]]) for r in range(n_reads)]
    with chaindp.Device(0, max_anchors=1 << 21, max_reads=64) as d:
        ix = d.load_index(blobs)
        import time
        for _ in range(2):
            t0 = time.time()
            off, a, rep_len, mpo, mp = d.collect_seeds(ix, flag, max_occ, np.array(mini_off, np.int64), mini, np.array(bid, np.uint32), np.array(qlen, np.int32))
            dt = time.time() - t0
    sizes = [len(e[0]) for e in exp]
    assert list(np.diff(off)) == sizes, (sizes, list(np.diff(off)))
    for r in range(n_reads):
        assert np.array_equal(a[off[r]:off[r + 1]], exp[r][0]), (seed, r, sizes[r], "anchors")
        assert rep_len[r] == exp[r][1] and np.array_equal(mp[mpo[r]:mpo[r + 1]], exp[r][2]), (seed, r)
    ties = sum(int((np.diff(e[0][:, 0]) == 0).sum()) for e in exp if len(e[0]) > 1)
    if not quiet:
        print(f"\nseed {seed}: anchors per read {sizes}, equal-x pairs {ties}, {dt * 1e3:.1f} ms with transfers")
    return sizes, ties


@pytest.mark.parametrize("seed,max_cnt,n_mini,rep_pct", [(1, 3, 400, 0), (2, 6, 900, 20), (3, 40, 500, 30), (4, 12, 2500, 40), (5, 90, 400, 25), (6, 90, 1500, 35),
                                                         (7, 90, 9000, 30)])
def test_gpu_seed_collection_against_host_statement_on_synthetic_images(seed, max_cnt, n_mini, rep_pct):
    """Random index images and minimizer lists with many repeated minimizers (equal x in the anchors) and reads of a few
    hundred to > 200 k anchors: every sort path (bitonic, the reference's procedure in LDS with 32 / 4 bucket tables, and for
    reads beyond the LDS sort the top levels by k_seed_sort_huge with digits in LDS or, past ~150 k anchors, in global
    memory).  Expected values come from the oracle's restatement (oracle/seed_oracle.cpp), itself pinned against the reference
    on the CPU tier."""
    _synthetic_case(seed, max_cnt, n_mini, rep_pct)


@pytest.mark.parametrize("seed,n_mini,rid_pool", [(8, 1050, 160), (9, 1100, 90), (10, 700, 60)])
def test_gpu_seed_collection_with_few_targets(seed, n_mini, rid_pool):
    """Reads of 8-13 k anchors on a few dozen targets: in the middle levels of the sort more than a hundred buckets of more
    than 64 anchors are pending at a time (the large-range slots of the LDS queues), with ties throughout."""
    _synthetic_case(seed, 40, n_mini, 30, rid_pool=rid_pool)


def test_gpu_seed_collection_fuzz():
    """Random shapes of the sort's input: one target to a million (how many buckets a level has), reference positions
    confined to 4..21 bits (on which levels equal keys separate, if at all), few to many hits per minimizer, reads of a
    few hundred to ~20 k anchors.  CHAINDP_FUZZ_SCALE multiplies the number of cases."""
    n_cases = 24 * max(1, int(os.environ.get("CHAINDP_FUZZ_SCALE", "1")))
    rng = np.random.default_rng(20260104)
    seen = []
    for c in range(n_cases):
        rid_pool = int(rng.choice([1, 2, 3, 7, 40, 150, 300, 700, 70000, 1 << 20]))
        pos_bits = int(rng.choice([4, 8, 9, 13, 17, 21]))
        max_cnt = int(rng.choice([3, 10, 40, 90]))
        n_mini = int(rng.choice([200, 600, 1500])) * (1 if max_cnt > 10 else 4)
        sizes, ties = _synthetic_case(1000 + c, max_cnt, n_mini, int(rng.choice([0, 20, 45])), rid_pool=rid_pool, pos_bits=pos_bits, quiet=True)
        seen.append((max(sizes), ties))
    print(f"\n{n_cases} cases; largest read {max(s for s, _ in seen)} anchors; cases with ties {sum(1 for _, t in seen if t)}")


@pytest.mark.parametrize("limits,lab_cap", [("256,512", "1024"), ("128,128", "100000"), ("1024,4096", "4096")])
@pytest.mark.parametrize("seed,max_cnt,n_mini,rep_pct", [(2, 6, 900, 20), (4, 12, 2500, 40), (6, 90, 1500, 35)])
def test_gpu_seed_collection_with_small_sort_limits(monkeypatch, limits, lab_cap, seed, max_cnt, n_mini, rep_pct):
    """The same cases with the LDS sort's limits turned down (test switches read when a context first collects seeds), so
    that ordinary reads take the paths made for very large ones: top levels by k_seed_sort_huge (digits in LDS and in
    global memory, nested large buckets, single-digit levels), buckets handed to both LDS configurations as work items,
    and, beyond 320 x the limit, the one-thread kernel."""
    monkeypatch.setenv("CHAINDP_SEED_MAX_N", limits)
    monkeypatch.setenv("CHAINDP_SEED_LAB_CAP", lab_cap)
    _synthetic_case(seed, max_cnt, n_mini, rep_pct)


def test_seed_collection_argument_errors(dev):
    g = np.load(SEEDS[0], allow_pickle=False)
    L, ctx = dev._lib, dev._ctx
    ix = dev.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    import ctypes as C
    off = np.zeros(2, np.int64); rep = np.zeros(1, np.int32); mpo = np.zeros(2, np.int64)
    mini_off = np.array([0, 3], np.int64); mini = np.zeros((3, 2), np.uint64); bid = np.zeros(1, np.uint32); qlen = np.full(1, 100, np.int32)
    args = lambda **kw: [kw.get("ctx", ctx), kw.get("ix", ix), 0, 50, kw.get("n", 1), kw.get("mo", mini_off.ctypes.data), kw.get("mi", mini.ctypes.data),
                         kw.get("bid", bid.ctypes.data), qlen.ctypes.data, None, off.ctypes.data, rep.ctypes.data, mpo.ctypes.data]
    assert L.chaindp_collect_seeds(*args()) == 0
    assert L.chaindp_collect_seeds(*args(ix=None)) != 0                       # no index image
    assert L.chaindp_collect_seeds(*args(mo=None)) != 0                       # no offsets
    assert L.chaindp_collect_seeds(*args(mi=None)) != 0                       # minimizers announced but absent
    assert L.chaindp_collect_seeds(*args(bid=None)) != 0
    assert L.chaindp_collect_seeds(*args(n=-1)) != 0
    assert b"index" in L.chaindp_last_error(ctx) or len(L.chaindp_last_error(ctx)) > 0
    # an image without its B blob is refused at creation
    assert not L.chaindp_index_create(0, None, 0, g["img_H"].ctypes.data, g["img_H"].size, g["img_V"].ctypes.data, g["img_V"].size, None, 0)
    # scatter without a matching seed collection
    dst = (C.c_void_p * 4)()
    assert L.chaindp_scatter_mini_pos(ctx, 4, dst) != 0
