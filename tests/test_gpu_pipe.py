"""GPU tier: the streaming pipe (chaindp_pipe_t, include/chaindp.h) against the synchronous path and the oracle.
Batches of different shapes (ragged, empty, single read) stream through three contexts; every batch's new_seed[]
must equal, byte for byte, what chaindp_compact returns for the same batch and what the oracle's compaction
(chain.c:286-317) produces."""
import numpy as np
import pytest

import oracle_lib as ol
from minimap2_chaindp_amd import anchorgen as ag, chaindp, params as P

pytestmark = pytest.mark.gpu


def _batches():
    out = []
    for k, (gen, n, over) in enumerate([("ava-ont", 60, {}), ("map-ont", 40, {}), ("ties", 30, {}), ("ava-ont", 1, {}),
                                        ("skew", 25, dict(skew_max=20000)), ("ava-ont", 90, {}), ("dense", 3, dict(read_len=2500, n_hits=10))]):
        off, a = ag.generate(gen, n_reads=n, seed=500 + k, **over)
        out.append((off, a))
    out.insert(3, (np.zeros(1, np.int64), np.zeros((0, 2), np.uint64)))          # an empty batch in the middle of the stream
    return out


@pytest.mark.parametrize("depth", [1, 3])
def test_pipe_matches_sync_path_and_oracle(depth):
    par = P.preset("ava-ont")
    batches = _batches()
    cap = max(int(o[-1]) for o, _ in batches) + 1
    with chaindp.Device(0, max_anchors=cap, max_reads=256) as dev, chaindp.Pipe(0, depth=depth, max_anchors=cap, max_reads=256) as pipe:
        want = []
        for off, a in batches:
            if int(off[-1]) == 0:
                want.append((np.zeros(len(off), np.int64), b""))
                continue
            dev.chain_batch(par, off, a)
            soff, seeds = dev.compact(par)
            want.append((soff, seeds.tobytes()))
        got = {}
        sub = 0
        while len(got) < len(batches):
            while sub < len(batches) and pipe.submit(par, batches[sub][0], batches[sub][1], tag=sub):
                sub += 1
            tag, soff, seeds = pipe.wait()
            pipe.release()
            got[tag] = (soff, seeds.tobytes())
        assert sub == len(batches)
        for k in range(len(batches)):
            assert np.array_equal(got[k][0], want[k][0]), (k, "seeds_off")
            assert got[k][1] == want[k][1], (k, "new_seed[] bytes")
    # and against the oracle's compaction for one of them
    off, a = batches[0]
    f, p, v, _ = ol.oracle_batch(par, off, a)
    recs = b"".join(ol.oracle_compact(par, a[int(off[r]):int(off[r + 1])], f[int(off[r]):int(off[r + 1])], p[int(off[r]):int(off[r + 1])],
                                      v[int(off[r]):int(off[r + 1])]).tobytes() for r in range(len(off) - 1))
    assert got[0][1] == recs


def test_pipe_busy_and_errors():
    par = P.preset("ava-ont")
    off, a = ag.generate("ava-ont", n_reads=5, seed=3)
    with chaindp.Pipe(0, depth=2, max_anchors=int(off[-1]) + 1, max_reads=16) as pipe:
        assert pipe.submit(par, off, a, tag=1)
        assert pipe.submit(par, off, a, tag=2)
        assert not pipe.submit(par, off, a, tag=3)               # both slots in flight: CHAINDP_ERR_BUSY
        t1, s1, r1 = pipe.wait()
        with pytest.raises(chaindp.ChainDPError):
            pipe.wait()                                          # the waited batch has to be released first
        pipe.release()
        t2, s2, r2 = pipe.wait()
        pipe.release()
        assert (t1, t2) == (1, 2) and np.array_equal(s1, s2) and r1.tobytes() == r2.tobytes()
        with pytest.raises(chaindp.ChainDPError):
            pipe.wait()                                          # nothing in flight
        big_off, big_a = ag.generate("ava-ont", n_reads=40, seed=4)
        with pytest.raises(chaindp.ChainDPError):
            pipe.submit(par, big_off, big_a, tag=9)              # over capacity
