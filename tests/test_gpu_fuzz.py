"""GPU tier: randomised parity.  Random DP parameters (gaps, bandwidth, max_skip, min_sc, cDNA, segments) on random
generator shapes, both kernel variants and all ring sizes, f/p/v and new_seed[] against the oracle, chains against
its bottom half.  Seeds are fixed, so a failure is reproducible from the printed case."""
import numpy as np
import pytest

import oracle_lib as ol
from minimap2_chaindp_amd import anchorgen as ag, chaindp, params as P

pytestmark = pytest.mark.gpu


def _cases(n):
    rng = np.random.default_rng(20261004)
    out = []
    for k in range(n):
        n_segs = int(rng.choice([1, 1, 1, 2, 3]))
        par = dict(max_dist_x=int(rng.choice([40, 300, 5000, 10000, 200000])), max_dist_y=int(rng.choice([30, 500, 5000, 10000])),
                   bw=int(rng.choice([0, 7, 100, 500, 2000, 5000])), max_skip=int(rng.choice([0, 1, 5, 25, 60])),
                   min_sc=int(rng.choice([0, 15, 40, 100])), is_cdna=int(rng.random() < 0.25), n_segs=n_segs)
        gen = dict(read_len=int(rng.choice([200, 1500, 6000])), n_hits=int(rng.integers(1, 25)), min_ovl_pct=int(rng.choice([10, 50, 90])),
                   step=int(rng.choice([2, 5, 20, 60])), indel_pct=int(rng.choice([0, 30, 80])), indel_max=int(rng.choice([1, 6, 40])),
                   noise_pct=int(rng.choice([0, 8, 40])), tie_pct=int(rng.choice([0, 2, 30])), q_span=int(rng.choice([1, 15, 28, 200])),
                   span_jitter=int(rng.choice([0, 9, 55])), n_ref=int(rng.choice([1, 4, 1000])), ref_len=int(rng.choice([3000, 20000])),
                   n_segs=n_segs)
        out.append((k, par, gen, int(rng.choice([128, 256, 512])), bool(rng.random() < 0.3), int(rng.integers(5, 40)), int(rng.choice([1, 2, 3]))))
    return out


def _fast_cases(n):
    """Cases the table-driven kernel variant takes (one segment, not cDNA, bw within the table), which is what ordinary
    long-read input runs on: more of them, with gaps and bandwidths around each other."""
    rng = np.random.default_rng(20261005)
    out = []
    for k in range(n):
        par = dict(max_dist_x=int(rng.choice([1, 40, 300, 2000, 5000, 10000, 100000])), max_dist_y=int(rng.choice([0, 30, 300, 2000, 5000, 10000, 50000])),
                   bw=int(rng.choice([0, 7, 100, 500, 2000, 4095])), max_skip=int(rng.choice([-1, 0, 1, 5, 25, 60, 200])),
                   min_sc=int(rng.choice([0, 15, 40, 100])), is_cdna=0, n_segs=1)
        gen = dict(read_len=int(rng.choice([200, 1500, 6000])), n_hits=int(rng.integers(1, 25)), min_ovl_pct=int(rng.choice([10, 50, 90])),
                   step=int(rng.choice([2, 5, 20, 60])), indel_pct=int(rng.choice([0, 30, 80])), indel_max=int(rng.choice([1, 6, 40])),
                   noise_pct=int(rng.choice([0, 8, 40])), tie_pct=int(rng.choice([0, 2, 30])), q_span=int(rng.choice([1, 15, 28, 200])),
                   span_jitter=int(rng.choice([0, 9, 55])), n_ref=int(rng.choice([1, 4, 1000])), ref_len=int(rng.choice([3000, 20000])),
                   n_segs=1)
        out.append((1000 + k, par, gen, int(rng.choice([128, 128, 256, 512])), False, int(rng.integers(5, 40)), int(rng.choice([1, 2, 3]))))
    return out


@pytest.fixture(scope="module")
def dev():
    with chaindp.Device(0, max_anchors=1 << 22, max_reads=1 << 12) as d:
        yield d


_SCALE = int(__import__("os").environ.get("CHAINDP_FUZZ_SCALE", "1"))     # more of the same, for an occasional long run


@pytest.mark.parametrize("k,par_kw,gen_kw,ring,general,n_reads,min_cnt", _cases(48 * _SCALE) + _fast_cases(64 * _SCALE))
def test_random_case(dev, k, par_kw, gen_kw, ring, general, n_reads, min_cnt):
    par = P.ChainParams(**par_kw)
    base = dict(ag.PRESETS["ties"]); base.update(gen_kw)
    off, a = ag.generate("ties", n_reads=n_reads, seed=1000 + k, threads=4, **gen_kw)
    while off[-1] > dev.max_anchors and n_reads > 1:          # a dense draw can exceed the fixture's capacity: keep fewer reads
        n_reads //= 2
        off, a = off[:n_reads + 1], a[:off[n_reads]]
    n_segs = None
    if par.n_segs > 1 and k % 2:                      # per-read n_segs as in collect_task_t
        n_segs = (np.arange(n_reads) % par.n_segs + 1).astype(np.int32)
    # general: k_chain_units' 64-bit variant; otherwise the table-driven code, alternately two units per wave (+ hand-over) and one
    dev.set_ring(ring); dev.set_variant(1 if general else (2 if k % 2 else 0))
    try:
        f, p, v = dev.chain_batch(par, off, a, n_segs=n_segs)
        of, op, ov, _ = ol.oracle_batch(par, off, a, n_segs=n_segs, threads=4)
        assert np.array_equal(f, of) and np.array_equal(p, op) and np.array_equal(v, ov), (k, par_kw, gen_kw, ring, general)
        soff, seeds = dev.compact(par)
        coff, u, boff, b = dev.backtrack(par, min_cnt)
        for r in range(n_reads):
            lo, hi = int(off[r]), int(off[r + 1])
            rp = P.ChainParams(**par_kw)
            if n_segs is not None:
                rp.n_segs = int(n_segs[r])
            exp = ol.oracle_compact(rp, np.ascontiguousarray(a[lo:hi]), of[lo:hi].copy(), op[lo:hi].copy(), ov[lo:hi].copy())
            got = seeds[int(soff[r]):int(soff[r + 1])]
            assert got.tobytes() == exp.tobytes(), (k, r, "new_seed")
            eu, eb = ol.oracle_bottom(min_cnt, par.min_sc, got)
            assert np.array_equal(u[int(coff[r]):int(coff[r + 1])], eu), (k, r, "u")
            assert np.array_equal(b[int(boff[r]):int(boff[r + 1])], eb.reshape(-1, 2)), (k, r, "b")
    finally:
        dev.set_ring(128); dev.set_variant(False)
