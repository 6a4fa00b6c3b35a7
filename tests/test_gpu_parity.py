"""GPU tier: the HIP path, called through the C ABI (libchaindp_hip.so), against the golden vectors
of the reference and against the oracle on seeded batches.  Bit-exact: integer arrays are compared
with array_equal, new_seed[] by bytes."""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import golden_names, load_golden, params_from
from minimap2_chaindp_amd import anchorgen as ag, chaindp, params as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    with chaindp.Device(0, max_anchors=1 << 23, max_reads=1 << 16) as d:
        yield d


@pytest.mark.parametrize("ring,general", [(128, 0), (256, 0), (512, 0), (128, 1), (256, 1), (128, 2), (256, 2), (512, 2)])
@pytest.mark.parametrize("name", golden_names())
def test_golden_fixture_bit_exact(dev, name, ring, general):
    """Every fixture through the three ways the DP can run -- 0: two units per wave (k_chain_twin) with k_chain_units for what it
    hands over, 1: k_chain_units' general 64-bit variant, 2: k_chain_units alone, table-driven where it applies -- and every
    LDS ring size of k_chain_units (small rings push the long scans onto the deep, global-memory path)."""
    g = load_golden(name)
    par = params_from(g["params"])
    dev.set_ring(ring)
    dev.set_variant(general)
    f, p, v = dev.chain_batch(par, g["off"], g["anchors"])
    assert np.array_equal(f, g["f"]), (name, "f", int(np.flatnonzero(f != g["f"])[0]))
    assert np.array_equal(p, g["p"]), (name, "p", int(np.flatnonzero(p != g["p"])[0]))
    assert np.array_equal(v, g["v"]), (name, "v", int(np.flatnonzero(v != g["v"])[0]))
    soff, seeds = dev.compact(par)
    assert np.array_equal(soff, g["seeds_off"]), (name, "new_i")
    assert seeds.tobytes() == g["seeds"].tobytes(), (name, "new_seed[] bytes")


CASES = [  # generator, generator overrides, DP preset, DP overrides, reads, per-read n_segs?
    ("ava-ont", {}, "ava-ont", {}, 300, False),
    ("map-ont", {}, "map-ont", {}, 200, False),
    ("ties", {}, "map-ont", {}, 200, False),
    ("ties", {}, "splice", {}, 100, False),
    ("ties", {}, "ava-pb", {}, 100, False),
    ("paired", {}, "sr", {}, 2000, False),
    ("paired", {}, "map-ont", dict(n_segs=2), 500, True),
    ("dense", dict(read_len=3000, n_hits=12), "ava-ont", {}, 6, False),
    ("skew", dict(skew_max=30000), "ava-ont", {}, 100, False),
    ("ties", {}, "map-ont", dict(max_skip=0), 50, False),
    ("ties", {}, "map-ont", dict(max_skip=2, bw=30), 50, False),
    ("ties", {}, "map-ont", dict(max_skip=-1), 30, False),            # ++n_skip > -1: the first marked predecessor breaks
    ("ties", {}, "map-ont", dict(min_sc=-5, bw=0), 30, False),        # everything is "v >= min_sc"; only the exact diagonal chains
    ("ava-ont", dict(q_span=255, span_jitter=0), "ava-ont", {}, 40, False),   # 8-bit span at its maximum
]


@pytest.mark.parametrize("variant", [0, 2], ids=["two_per_wave", "one_per_wave"])
@pytest.mark.parametrize("gen,gen_over,preset,par_over,n_reads,per_read", CASES)
def test_seeded_batches_match_oracle(dev, gen, gen_over, preset, par_over, n_reads, per_read, variant):
    dev.set_ring(128)
    dev.set_variant(variant)
    par = P.preset(preset, **par_over)
    off, a = ag.generate(gen, n_reads=n_reads, seed=1234, **gen_over)
    n_segs = None
    if per_read:   # collect_task_t::n_segs is per read (fpga_chaindp.h:53): mix 1 and 2
        n_segs = (np.arange(n_reads) % 2 + 1).astype(np.int32)
    f, p, v = dev.chain_batch(par, off, a, n_segs=n_segs)
    of, op, ov, _ = ol.oracle_batch(par, off, a, n_segs=n_segs, threads=8)
    for name, x, y in (("f", f, of), ("p", p, op), ("v", v, ov)):
        bad = np.flatnonzero(x != y)
        assert bad.size == 0, (gen, preset, name, "first mismatch at anchor", int(bad[0]), int(x[bad[0]]), int(y[bad[0]]),
                               "read", int(np.searchsorted(off, bad[0], side="right") - 1))
    soff, seeds = dev.compact(par)
    assert int(soff[-1]) == len(seeds)
    for r in range(0, n_reads, max(1, n_reads // 25)):
        lo, hi = int(off[r]), int(off[r + 1])
        rp = P.preset(preset, **par_over)
        if n_segs is not None:
            rp.n_segs = int(n_segs[r])
        exp = ol.oracle_compact(rp, np.ascontiguousarray(a[lo:hi]), of[lo:hi].copy(), op[lo:hi].copy(), ov[lo:hi].copy())
        assert seeds[int(soff[r]):int(soff[r + 1])].tobytes() == exp.tobytes(), (gen, preset, r)


@pytest.mark.parametrize("variant", [0, 2], ids=["two_per_wave", "one_per_wave"])
def test_zero_and_mixed_q_spans(dev, variant):
    """k_chain_twin keeps scores minus one with a floor of zero, which is only right while q_span >= 1: a read with a zero q_span
    must leave it for k_chain_units (SUMQ_SPAN0_FLAG, set by the prepass), the other reads of the same batch must not, and reads
    whose spans differ (another avg_qspan: another cost table) must reload the table when a half moves from one to the next."""
    dev.set_ring(128)
    dev.set_variant(variant)
    par = P.preset("ava-ont")
    off, a = ag.generate("ava-ont", n_reads=60, seed=77)
    a = a.copy()
    rng = np.random.default_rng(5)
    for r in range(60):
        lo, hi = int(off[r]), int(off[r + 1])
        if r % 3 == 0:                                     # every third read: a few anchors with q_span 0
            idx = lo + rng.choice(hi - lo, size=max(1, (hi - lo) // 50), replace=False)
            a[idx, 1] &= ~np.uint64(0xff << 32)
        elif r % 3 == 1:                                   # every third read: its own span (1..40), so its own avg_qspan and table
            a[lo:hi, 1] = (a[lo:hi, 1] & ~np.uint64(0xff << 32)) | np.uint64((1 + r % 40) << 32)
    f, p, v = dev.chain_batch(par, off, a)
    of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
    for name, x, y in (("f", f, of), ("p", p, op), ("v", v, ov)):
        bad = np.flatnonzero(x != y)
        assert bad.size == 0, (name, "first mismatch at anchor", int(bad[0]), int(x[bad[0]]), int(y[bad[0]]), "read", int(np.searchsorted(off, bad[0], side="right") - 1))
    soff, seeds = dev.compact(par)
    for r in range(0, 60, 5):
        lo, hi = int(off[r]), int(off[r + 1])
        exp = ol.oracle_compact(par, np.ascontiguousarray(a[lo:hi]), of[lo:hi].copy(), op[lo:hi].copy(), ov[lo:hi].copy())
        assert seeds[int(soff[r]):int(soff[r + 1])].tobytes() == exp.tobytes(), r
    if variant == 0:
        assert dev.leftover_units() > 0                    # the reads with a zero span went to k_chain_units


@pytest.mark.parametrize("mode", [1, 2], ids=["untouched", "resumed_after_first_tile"])
@pytest.mark.parametrize("gen,preset,n_reads", [("ava-ont", "ava-ont", 200), ("map-ont", "map-ont", 60), ("ties", "map-ont", 120), ("skew", "ava-ont", 60)])
def test_units_handed_over_by_the_twin_kernel(dev, gen, preset, n_reads, mode):
    """What k_chain_twin hands to k_chain_units: a unit it never touched is done from scratch, a unit it gave up on (a scan past its
    ring, too many second chunks) is RESUMED behind the tiles it had flushed -- the ring rebuilt from a, f, p, v.  The test hook
    sends every unit down one road or the other; results must not depend on who scored which tile."""
    dev.set_ring(128)
    dev.set_variant(0)
    dev.set_twin_handover(mode)
    try:
        par = P.preset(preset)
        off, a = ag.generate(gen, n_reads=n_reads, seed=2024)
        f, p, v = dev.chain_batch(par, off, a)
        left = dev.leftover_units()
        of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
        for name, x, y in (("f", f, of), ("p", p, op), ("v", v, ov)):
            bad = np.flatnonzero(x != y)
            assert bad.size == 0, (gen, mode, name, "first mismatch at anchor", int(bad[0]), int(x[bad[0]]), int(y[bad[0]]))
        assert left > 0, (gen, mode)
        soff, seeds = dev.compact(par)
        for r in range(0, n_reads, max(1, n_reads // 20)):
            lo, hi = int(off[r]), int(off[r + 1])
            exp = ol.oracle_compact(par, np.ascontiguousarray(a[lo:hi]), of[lo:hi].copy(), op[lo:hi].copy(), ov[lo:hi].copy())
            assert seeds[int(soff[r]):int(soff[r + 1])].tobytes() == exp.tobytes(), (gen, mode, r)
    finally:
        dev.set_twin_handover(0)


@pytest.mark.parametrize("mode", [0, 2], ids=["plain", "handover_after_first_tile"])
@pytest.mark.parametrize("gen,gen_over,preset,n_reads", [("ava-ont", {}, "ava-ont", 300), ("skew", dict(skew_max=30000), "ava-ont", 100),
                                                        ("ava-ont", dict(q_span=21, span_jitter=0), "ava-ont", 40), ("ava-ont", dict(noise_pct=40, tie_pct=20), "ava-ont", 120)])
def test_four_units_per_wave_variant(dev, gen, gen_over, preset, n_reads, mode):
    """k_chain_quad (chaindp_quad.hip: four units per wave, two predecessors per lane; off by default because it measured slower than
    k_chain_twin) takes batches whose reads all have one cost table: same results, element for element -- also when every unit is
    handed to k_chain_units after its first tile, and on a noisy shape whose scans interleave new maxima and marked predecessors."""
    dev.set_ring(128)
    dev.set_variant(0)
    dev.set_quad(True)
    dev.set_twin_handover(mode)
    try:
        par = P.preset(preset)
        off, a = ag.generate(gen, n_reads=n_reads, seed=99, **gen_over)
        f, p, v = dev.chain_batch(par, off, a)
        assert dev.quad_took()
        of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
        for name, x, y in (("f", f, of), ("p", p, op), ("v", v, ov)):
            bad = np.flatnonzero(x != y)
            assert bad.size == 0, (gen, mode, name, "first mismatch at anchor", int(bad[0]), int(x[bad[0]]), int(y[bad[0]]))
        soff, seeds = dev.compact(par)
        for r in range(0, n_reads, max(1, n_reads // 15)):
            lo, hi = int(off[r]), int(off[r + 1])
            exp = ol.oracle_compact(par, np.ascontiguousarray(a[lo:hi]), of[lo:hi].copy(), op[lo:hi].copy(), ov[lo:hi].copy())
            assert seeds[int(soff[r]):int(soff[r + 1])].tobytes() == exp.tobytes(), (gen, mode, r)
    finally:
        dev.set_quad(False)
        dev.set_twin_handover(0)


def test_general_variant_on_seeded_batches(dev):
    dev.set_ring(256)
    dev.set_variant(True)
    try:
        for gen, preset, n_reads in (("ava-ont", "ava-ont", 100), ("dense", "ava-ont", 3), ("ties", "map-ont", 60)):
            par = P.preset(preset)
            off, a = ag.generate(gen, n_reads=n_reads, seed=4321)
            f, p, v = dev.chain_batch(par, off, a)
            of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
            assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov), (gen, preset)
    finally:
        dev.set_variant(False)


def test_large_bw_takes_the_general_path(dev):
    """bw above the cost table's limit (and a huge max_dist_x, where 32-bit ring arithmetic is not provably
    exact) must fall back to the general variant and still match."""
    dev.set_ring(128)
    par = P.preset("map-ont", bw=6000, max_dist_x=40_000_000, max_dist_y=20000)
    off, a = ag.generate("ties", n_reads=30, seed=9)
    f, p, v = dev.chain_batch(par, off, a)
    of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
    assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov)


@pytest.mark.parametrize("par_over", [
    dict(max_dist_x=10000, max_dist_y=3000),              # query gap below reference gap: saturating-add form of the range test
    dict(max_dist_x=3000, max_dist_y=10000),              # ... and above it
    dict(max_dist_x=300, max_dist_y=300, bw=500),         # bw >= max_dist_x: the bandwidth test can never fail on its own
    dict(max_dist_x=5000, max_dist_y=0),                  # nothing has 1 <= dq <= 0
    dict(max_dist_x=1, max_dist_y=1, bw=0),
    dict(max_skip=-1, max_dist_y=700),
])
@pytest.mark.parametrize("gen,n_reads", [("dense", 2), ("ties", 40)])
@pytest.mark.parametrize("ring,handover", [(128, False), (512, False), (128, 2), (256, 2), (128, 3), (128, 4)])
def test_fast_variant_parameter_corners(dev, par_over, gen, n_reads, ring, handover):
    """The table-driven variant folds the window, gap and bandwidth tests of chain.c:252-260 into one unsigned compare and keeps
    marks by distance; 'dense' walks the whole window (ring chunks, far marks, the deep path), 'ties' breaks early.  Without the
    handover k_chain_units serves the long scans of the dense units from HBM/L2 to the end; with it they are redone by
    k_chain_dense (eight waves per unit, marks as bits by distance in LDS; mode 2), k_chain_dense1 (one wave per unit, the same
    marks; mode 3) or k_chain_dense16 (sixteen waves per unit, rounds of 1024 predecessors; mode 4)."""
    dev.set_ring(ring)
    dev.set_variant(False)
    dev.set_deep_handover(handover)
    try:
        par = P.preset("ava-ont", **par_over)
        off, a = ag.generate(gen, n_reads=n_reads, seed=77, **(dict(read_len=2500, n_hits=10) if gen == "dense" else {}))
        f, p, v = dev.chain_batch(par, off, a)
        of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
        assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov), (par_over, gen, ring)
        soff, seeds = dev.compact(par)
        for r in range(n_reads):
            lo, hi = int(off[r]), int(off[r + 1])
            exp = ol.oracle_compact(par, np.ascontiguousarray(a[lo:hi]), of[lo:hi].copy(), op[lo:hi].copy(), ov[lo:hi].copy())
            assert seeds[int(soff[r]):int(soff[r + 1])].tobytes() == exp.tobytes(), (par_over, gen, r)
    finally:
        dev.set_deep_handover(True)
        dev.set_ring(128)


@pytest.mark.parametrize("mode", [True, 2, 3, 4], ids=["as_the_batch_decides", "eight_waves_per_unit", "one_wave_per_unit", "sixteen_waves_per_unit"])
@pytest.mark.parametrize("gen_kw", [dict(read_len=3000, n_hits=12), dict(read_len=4000, n_hits=20), dict(read_len=6000, n_hits=6)])
def test_dense_units_are_redone_by_the_dense_kernel(dev, gen_kw, mode):
    """Default settings: units whose scans keep reaching past the LDS ring (dense repeats) are handed by k_chain_units to one of the
    dense kernels, which redoes them from scratch -- same f/p/v and new_seed[] as the oracle.  A batch this small has a short tail:
    sixteen waves per unit (k_chain_dense16); mode 2 sends the units to k_chain_dense (eight waves: the kernel of a longer tail),
    mode 3 to k_chain_dense1 (one wave: batches that are dense all over), mode 4 to k_chain_dense16 whatever the batch looks like."""
    dev.set_ring(128)
    dev.set_deep_handover(mode)
    par = P.preset("ava-ont")
    off, a = ag.generate("dense", n_reads=3, seed=5, **gen_kw)
    of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
    try:
        for variant in (0, 2):
            dev.set_variant(variant)
            f, p, v = dev.chain_batch(par, off, a)
            assert dev.deep_units() > 0, variant
            assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov), variant
            soff, seeds = dev.compact(par)
            for r in range(len(off) - 1):
                lo, hi = int(off[r]), int(off[r + 1])
                exp = ol.oracle_compact(par, np.ascontiguousarray(a[lo:hi]), of[lo:hi].copy(), op[lo:hi].copy(), ov[lo:hi].copy())
                assert seeds[int(soff[r]):int(soff[r + 1])].tobytes() == exp.tobytes(), (variant, r)
    finally:
        dev.set_variant(0)
        dev.set_deep_handover(True)


def test_dense_unit_longer_than_the_mark_bitmap_stays_with_k_chain_units(dev):
    """k_chain_dense keeps marks as one bit per distance for 65536 distances; a unit with more anchors than that is not handed
    over, however deep its scans run, and k_chain_units serves it from HBM/L2."""
    dev.set_ring(128)
    par = P.preset("ava-ont")
    off, a = ag.generate("dense", n_reads=1, seed=9, read_len=14000, n_hits=28)        # units of 62 k and 98 k anchors
    x = a[:, 0]
    cuts = np.flatnonzero((x[1:] >> np.uint64(32)) != (x[:-1] >> np.uint64(32))) + 1
    lens = np.diff(np.concatenate(([0], cuts, [len(x)])))
    assert lens.max() > 65536 > lens.min()
    of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
    f, p, v = dev.chain_batch(par, off, a)
    assert dev.deep_units() == 1
    assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov)


def test_reference_position_crossing_2_to_32(dev):
    """x = rid << 32 | pos is compared as a 64-bit number (chain.c:252); the table-driven variant keeps only x.lo in LDS and
    relies on modular differences.  A unit whose x.lo wraps around 2^32 must still match."""
    rng = np.random.default_rng(5)
    n = 3000
    x = (np.uint64((1 << 32) - 6000) + np.cumsum(rng.integers(0, 9, n)).astype(np.uint64))
    assert x[0] < (1 << 32) <= x[-1]
    q = (np.cumsum(rng.integers(0, 9, n)) + 50).astype(np.uint64)
    a = np.stack([x, (np.uint64(15) << np.uint64(32)) | q], 1)
    off = np.array([0, n], np.int64)
    for ring in (128, 256):
        dev.set_ring(ring)
        for preset in ("map-ont", "ava-ont"):
            par = P.preset(preset)
            f, p, v = dev.chain_batch(par, off, a)
            of, op, ov, _ = ol.oracle_fpv(par, a)
            assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov), (ring, preset)
    dev.set_ring(128)


def test_edge_batches(dev):
    dev.set_ring(128)
    par = P.preset("map-ont")
    # empty batch, batch of empty reads, a single one-anchor read
    f, p, v = dev.chain_batch(par, np.zeros(1, np.int64), np.zeros((0, 2), np.uint64))
    assert len(f) == 0
    f, p, v = dev.chain_batch(par, np.zeros(4, np.int64), np.zeros((0, 2), np.uint64))
    assert len(f) == 0
    soff, seeds = dev.compact(par)
    assert list(soff) == [0, 0, 0, 0] and len(seeds) == 0
    a = np.array([[10, (15 << 32) | 20]], np.uint64)
    f, p, v = dev.chain_batch(par, np.array([0, 1], np.int64), a)
    assert (int(f[0]), int(p[0]), int(v[0])) == (15, -1, 15)


def test_big_span_sum_rounding(dev):
    """avg_qspan = (float)sum_qspan / n with sum_qspan > 2^24: the u64 -> f32 conversion must round
    like the host's (chain.c:241).  One read of 120k anchors with span 255/254 mixed."""
    n = 120_000
    rng = np.random.default_rng(1)
    span = np.where(rng.random(n) < 0.37, 254, 255).astype(np.uint64)
    x = np.cumsum(rng.integers(1, 9, n)).astype(np.uint64)
    q = (np.cumsum(rng.integers(1, 9, n)) + 300).astype(np.uint64)
    a = np.stack([x, (span << np.uint64(32)) | q], 1)
    off = np.array([0, n], np.int64)
    par = P.preset("map-ont")
    f, p, v = dev.chain_batch(par, off, a)
    of, op, ov, _ = ol.oracle_fpv(par, a)
    assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov)


def test_rejects_bad_arguments(dev):
    par = P.preset("map-ont", max_dist_x=-1)
    with pytest.raises(chaindp.ChainDPError):
        dev.chain_batch(par, np.array([0, 1], np.int64), np.array([[1, 1]], np.uint64))
    with pytest.raises(chaindp.ChainDPError):   # capacity
        small = chaindp.Device(0, max_anchors=10, max_reads=2)
        try:
            small.chain_batch(P.preset("map-ont"), np.array([0, 20], np.int64), np.zeros((20, 2), np.uint64))
        finally:
            small.close()


def test_run_is_idempotent_and_staged_api(dev):
    par = P.preset("ava-ont")
    off, a = ag.generate("ava-ont", n_reads=50, seed=77)
    dev.set_ring(128)
    dev.upload(off, a)
    dev.run(par); dev.sync()
    f1, p1, v1 = dev.download()
    dev.run(par); dev.run(par); dev.sync()
    f2, p2, v2 = dev.download()
    assert np.array_equal(f1, f2) and np.array_equal(p1, p2) and np.array_equal(v1, v2)
    st = dev.stats()
    assert st["anchors"] == int(off[-1]) and st["reads"] == 50 and st["units"] > 0


@pytest.mark.parametrize("min_sc", [40, 10], ids=["singletons_dropped", "singletons_emitted"])
def test_compaction_works_from_the_singleton_masks(dev, min_sc):
    """The prepass only MARKS singletons (anchors with nothing in reach on either side); their f, p, v and flag byte are written on
    demand (chaindp_download, chaindp_run_device).  Compaction straight after the run -- those arrays still holding another batch's
    values at the singletons' places -- must give the reference's new_seed[], with min_sc above q_span (no singleton is emitted) and
    below it (every singleton is, chain.c:304); the download afterwards must give the reference's f, p, v."""
    dev.set_ring(128); dev.set_variant(0)
    stale_off, stale_a = ag.generate("map-ont", n_reads=30, seed=5)
    dev.chain_batch(P.preset("map-ont"), stale_off, stale_a)
    dev.compact(P.preset("map-ont"))
    par = P.preset("ava-ont", min_sc=min_sc)
    off, a = ag.generate("ava-ont", n_reads=60, seed=99)
    dev.upload(off, a)
    dev.run(par)
    soff, seeds = dev.compact(par)
    of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
    assert int(soff[-1]) == len(seeds)
    n_single_records = 0
    for r in range(len(off) - 1):
        lo, hi = int(off[r]), int(off[r + 1])
        exp = ol.oracle_compact(par, np.ascontiguousarray(a[lo:hi]), of[lo:hi].copy(), op[lo:hi].copy(), ov[lo:hi].copy())
        assert seeds[int(soff[r]):int(soff[r + 1])].tobytes() == exp.tobytes(), (min_sc, r)
    single = (op < 0) & (ov == of) & (of == ((a[:, 1] >> np.uint64(32)) & np.uint64(0xff)).astype(np.int32))
    assert single.sum() > 100                                           # the batch has them
    n_single_records = int((single & (ov >= min_sc)).sum())
    assert (n_single_records > 0) == (min_sc <= 15)
    f, p, v = dev.download()
    assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov)


def test_run_on_the_callers_device_arrays(dev):
    """chaindp_run_device: offsets, anchors and the three result arrays are the caller's own HBM allocations (plain hipMalloc here);
    the results, singletons included, are complete when the stream has drained."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    H2D, D2H = 1, 2

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        return p

    dev.set_ring(128); dev.set_variant(0)
    par = P.preset("ava-ont")
    off, a = ag.generate("ava-ont", n_reads=40, seed=31)
    off = np.ascontiguousarray(off, np.int64); a = np.ascontiguousarray(a, np.uint64)
    n = int(off[-1])
    d_off, d_a = dmalloc(off.nbytes), dmalloc(a.nbytes)
    d_res = [dmalloc(4 * n) for _ in range(3)]
    try:
        assert hip.hipMemcpy(d_off, off.ctypes.data, off.nbytes, H2D) == 0 and hip.hipMemcpy(d_a, a.ctypes.data, a.nbytes, H2D) == 0
        for d in d_res:
            assert hip.hipMemset(d, 0x5a, 4 * n) == 0
        dev.run_device(par, len(off) - 1, n, d_off.value, d_a.value, None, d_res[0].value, d_res[1].value, d_res[2].value)
        dev.sync()
        got = [np.empty(n, np.int32) for _ in range(3)]
        for g, d in zip(got, d_res):
            assert hip.hipMemcpy(g.ctypes.data, d, 4 * n, D2H) == 0
    finally:
        for d in [d_off, d_a] + d_res:
            hip.hipFree(d)
    of, op, ov, _ = ol.oracle_batch(par, off, a, threads=8)
    assert np.array_equal(got[0], of) and np.array_equal(got[1], op) and np.array_equal(got[2], ov)


def test_reference_anchor_dumps_chain_like_the_oracle(dev):
    """Real-read dumps (oracle/mt_dump.c: the reference's own front half on its test/*.fa) through
    the GPU, with each read's own DP arguments."""
    import os
    from minimap2_chaindp_amd import dump
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dumps")
    reads = []
    for fn in sorted(os.listdir(d)):
        reads += dump.read_dump(os.path.join(d, fn))
    assert len(reads) == 3
    dev.set_ring(128)
    for par, min_cnt, off, a, idx in dump.batches(reads):
        f, p, v = dev.chain_batch(par, off, a)
        of, op, ov, _ = ol.oracle_batch(par, off, a, threads=2)
        assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov)
        soff, seeds = dev.compact(par)
        coff, u, boff, b = dev.backtrack(par, min_cnt)
        for r in range(len(off) - 1):
            eu, eb = ol.oracle_bottom(min_cnt, par.min_sc, seeds[int(soff[r]):int(soff[r + 1])])
            assert np.array_equal(u[int(coff[r]):int(coff[r + 1])], eu) and np.array_equal(b[int(boff[r]):int(boff[r + 1])], eb.reshape(-1, 2))
    # the MT read is BASELINE config 1: one chain, score 3189, 342 anchors
    mt = [x for x in reads if x[2].shape[0] == 346][0]
    dev.chain_batch(mt[0], np.array([0, 346], np.int64), mt[2]); dev.compact(mt[0])
    coff, u, boff, b = dev.backtrack(mt[0], mt[1])
    assert len(u) == 1 and int(u[0]) >> 32 == 3189 and int(u[0]) & 0xffffffff == 342
