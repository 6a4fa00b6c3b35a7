"""GPU tier, boxes with more than one MI355X only: the packet ABI over every visible GPU.  fpga_init starts two service
contexts per GPU (reference fpga.h:45 hands a tid to the driver; this driver lets whichever context is free take the next
packets), so the node's GPUs share one packet stream.  Checked: every read comes back bit-exact whichever GPU chained it, every
GPU took part, and the anchors each one chained are within a third of an even split (the split follows the work they get
done, not a static deal).  Skipped on the one-GPU boxes this repository is developed on."""
import threading

import numpy as np
import pytest

import oracle_lib as ol
from minimap2_chaindp_amd import anchorgen as ag, chaindp, fpga, params as P

pytestmark = pytest.mark.gpu


def _stream_and_check(n_groups_cfg, n_expected, fair_split=True):
    n_gpus = n_expected
    par = P.preset("ava-ont")
    n_reads, per_packet, n_threads = 400 * n_gpus, 8, 8
    off, a = ag.generate("ava-ont", n_reads=n_reads, seed=4242)
    reads = [(r, a[off[r]:off[r + 1]]) for r in range(n_reads)]
    packets = [reads[k:k + per_packet] for k in range(0, n_reads, per_packet)]
    with fpga.Driver(bw=par.bw, is_cdna=par.is_cdna, max_skip=par.max_skip, min_sc=par.min_sc, max_packets_per_batch=4, n_groups=n_groups_cfg) as drv:
        def producer(tid):
            for k in range(tid, len(packets), n_threads):
                assert drv.submit(fpga.build_task_packet(packets[k], gap_ref=par.max_dist_x, gap_qry=par.max_dist_y, tid=tid), tid) == 0
        ths = [threading.Thread(target=producer, args=(t,)) for t in range(n_threads)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        results = drv.wait_results(len(packets))
        per_gpu = drv.stats_gpu()
        st = drv.stats()
    assert len(per_gpu) == n_gpus and st["anchors"] == int(off[-1]) and sum(x[1] for x in per_gpu) == int(off[-1])
    fair = int(off[-1]) / n_gpus
    assert all(x[0] > 0 and x[1] > 0 for x in per_gpu), per_gpu                      # every GPU (group) took part
    if fair_split:
        assert all(abs(x[1] - fair) <= fair / 3 for x in per_gpu), per_gpu
    seen = {}
    for raw in results:
        for read_id, err, seeds in fpga.parse_result_packet(raw):
            assert err == 0 and read_id not in seen
            seen[read_id] = seeds
    for r in range(0, n_reads, 7):                                     # every seventh read against the oracle
        ar = np.ascontiguousarray(a[off[r]:off[r + 1]])
        f, p, v, _ = ol.oracle_fpv(par, ar)
        assert seen[r].tobytes() == ol.oracle_compact(par, ar, f, p, v).tobytes(), r


@pytest.mark.skipif(chaindp.device_count() < 2, reason="needs at least two GPUs")
def test_packet_stream_is_shared_by_all_gpus_and_bit_exact():
    _stream_and_check(0, chaindp.device_count())


def test_two_service_groups_share_one_packet_stream_on_any_box():
    """The shim's multi-GPU dispatch with two service groups (each with its own two contexts and counters), on however many GPUs the
    box has -- on a one-GPU box both groups run on GPU 0: same packets, same checks as the multi-GPU test."""
    _stream_and_check(2, 2, fair_split=chaindp.device_count() >= 2)      # (two groups on one GPU take turns as the scheduler lets them)


@pytest.mark.skipif(chaindp.device_count() < 2, reason="needs at least two GPUs")
def test_contexts_on_two_gpus_agree():
    par = P.preset("map-ont")
    off, a = ag.generate("map-ont", n_reads=40, seed=8)
    out = []
    for d in range(2):
        with chaindp.Device(d, max_anchors=int(off[-1]) + 1, max_reads=64) as dev:
            f, p, v = dev.chain_batch(par, off, a)
            soff, seeds = dev.compact(par)
            out.append((f, p, v, soff, seeds.tobytes()))
    assert all(np.array_equal(x, y) for x, y in zip(out[0][:4], out[1][:4])) and out[0][4] == out[1][4]
