"""GPU tier: the reference's driver ABI (fpga.h) served by the GPU: anchor packets in, result packets
out, from several producer threads, checked against the oracle's new_seed[] byte for byte."""
import glob
import os
import threading

import numpy as np
import pytest

import oracle_lib as ol
from minimap2_chaindp_amd import anchorgen as ag, fpga, params as P

pytestmark = pytest.mark.gpu


def test_packets_through_driver_abi_match_oracle():
    par = P.preset("ava-ont")
    n_reads, per_packet, n_threads = 200, 8, 4                         # 8 reads per packet: map.c:423
    off, a = ag.generate("ava-ont", n_reads=n_reads, seed=31)
    reads = [(r, a[off[r]:off[r + 1]]) for r in range(n_reads)]
    packets = [reads[k:k + per_packet] for k in range(0, n_reads, per_packet)]
    with fpga.Driver(bw=par.bw, is_cdna=par.is_cdna, max_skip=par.max_skip, min_sc=par.min_sc,
                     max_packets_per_batch=7) as drv:
        def producer(tid):
            for k in range(tid, len(packets), n_threads):
                pkt = fpga.build_task_packet(packets[k], gap_ref=par.max_dist_x, gap_qry=par.max_dist_y, tid=tid)
                assert drv.submit(pkt, tid) == 0
        ths = [threading.Thread(target=producer, args=(t,)) for t in range(n_threads)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        results = drv.wait_results(len(packets))
        st = drv.stats()
    assert st["packets"] == len(packets) and st["reads"] == n_reads and st["anchors"] == int(off[-1]) and st["err_reads"] == 0
    seen = {}
    for raw in results:
        for read_id, err, seeds in fpga.parse_result_packet(raw):
            assert err == 0 and read_id not in seen
            seen[read_id] = seeds
    assert sorted(seen) == list(range(n_reads))                        # matched by read_id, any order (map.c:930)
    for r in range(n_reads):
        ar = np.ascontiguousarray(a[off[r]:off[r + 1]])
        f, p, v, _ = ol.oracle_fpv(par, ar)
        exp = ol.oracle_compact(par, ar, f, p, v)
        assert seen[r].tobytes() == exp.tobytes(), r


def test_minimizer_packets_get_err_flag_and_mixed_gaps_are_grouped():
    par = P.preset("map-ont")
    off, a = ag.generate("map-ont", n_reads=6, seed=5, read_len=2000)
    with fpga.Driver(bw=par.bw, is_cdna=0, max_skip=par.max_skip, min_sc=par.min_sc) as drv:
        # the reference's own packet type (minimizers, type 3) WITHOUT an index image: answered with err_flag=1, header only
        drv.submit(fpga.build_task_packet([(0, a[off[0]:off[1]])], 5000, 5000, pkt_type=fpga.PKT_MINIMIZERS))
        # two reads with different (gap_ref, gap_qry) in one packet stream
        drv.submit(fpga.build_task_packet([(1, a[off[1]:off[2]])], 5000, 5000))
        drv.submit(fpga.build_task_packet([(2, a[off[2]:off[3]])], 800, 600))
        results = drv.wait_results(3)
        st = drv.stats()
    got = {}
    for raw in results:
        for read_id, err, seeds in fpga.parse_result_packet(raw):
            got[read_id] = (err, seeds)
    assert got[0][0] == 1 and got[0][1] is None and st["err_reads"] == 1
    for rid, (gx, gy) in ((1, (5000, 5000)), (2, (800, 600))):
        pr = P.preset("map-ont", max_dist_x=gx, max_dist_y=gy)
        ar = np.ascontiguousarray(a[off[rid]:off[rid + 1]])
        f, p, v, _ = ol.oracle_fpv(pr, ar)
        assert got[rid][1].tobytes() == ol.oracle_compact(pr, ar, f, p, v).tobytes()


SEED_FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seeds", "*.npz")))


@pytest.mark.parametrize("path", SEED_FIXTURES, ids=[os.path.basename(p)[:-4] for p in SEED_FIXTURES])
def test_unmodified_minimizer_packets_end_to_end(path):
    """The reference's own packets (type 3, minimizers; map.c:286-324) with its index image loaded through
    fpga_load_index: seeds are collected by the shim's host threads, chained on the GPU, and every result packet carries
    what fpga_work (map.c:484-568) would have produced: new_seed[] of the reference's anchors, mini_pos[] and rep_len."""
    g = np.load(path, allow_pickle=False)
    pv = [int(x) for x in g["params"]]          # max_dist_x, max_dist_y, bw, max_skip, min_sc, is_cdna, n_segs, min_cnt
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    n_reads = len(g["bid"])
    reads = [(r, g["mini"][g["mini_off"][r]:g["mini_off"][r + 1]], int(g["bid"][r]), int(g["qlen"][r])) for r in range(n_reads)]
    with fpga.Driver(bw=par.bw, is_cdna=par.is_cdna, max_skip=par.max_skip, min_sc=par.min_sc, flag=int(g["flag"]),
                     max_occ=int(g["mid_occ"]), index=[g["img_B"], g["img_H"], g["img_V"], g["img_P"]]) as drv:
        n_pkts = 0
        for k in range(0, n_reads, 8):
            drv.submit(fpga.build_task_packet(reads[k:k + 8], par.max_dist_x, par.max_dist_y, pkt_type=fpga.PKT_MINIMIZERS), tid=k // 8)
            n_pkts += 1
        results = drv.wait_results(n_pkts)
        st = drv.stats()
    assert st["err_reads"] == 0 and st["reads"] == n_reads and st["anchors"] == len(g["anchors"])
    seen = {}
    for raw in results:
        for read_id, err, seeds, mini_pos, rep_len in fpga.parse_result_packet_full(raw):
            assert err == 0
            seen[read_id] = (seeds, mini_pos, rep_len)
    assert sorted(seen) == list(range(n_reads))
    for r in range(n_reads):
        a = np.ascontiguousarray(g["anchors"][g["a_off"][r]:g["a_off"][r + 1]])
        f, p, v, _ = ol.oracle_fpv(par, a)
        exp = ol.oracle_compact(par, a, f, p, v)
        assert seen[r][0].tobytes() == exp.tobytes(), (os.path.basename(path), r, "new_seed[]")
        assert np.array_equal(seen[r][1], g["mini_pos"][g["mp_off"][r]:g["mp_off"][r + 1]]), (r, "mini_pos")
        assert seen[r][2] == int(g["rep_len"][r]), (r, "rep_len")


def test_minimizer_and_anchor_packets_in_one_stream():
    """Both payload kinds submitted back to back (they share device batches, which then have several groups and assemble
    their result packets from staged copies): every read comes back right whichever way its seeds travelled."""
    path = [p for p in SEED_FIXTURES if "syn_repeats_mapont" in p][0]
    g = np.load(path, allow_pickle=False)
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    n = len(g["bid"])
    anchors = [np.ascontiguousarray(g["anchors"][g["a_off"][r]:g["a_off"][r + 1]]) for r in range(n)]
    with fpga.Driver(bw=par.bw, is_cdna=par.is_cdna, max_skip=par.max_skip, min_sc=par.min_sc, flag=int(g["flag"]),
                     max_occ=int(g["mid_occ"]), index=[g["img_B"], g["img_H"], g["img_V"], g["img_P"]]) as drv:
        pkts = []
        for r in range(n):                               # read r as minimizers (id r) and as anchors (id 1000 + r), alternating
            pkts.append(fpga.build_task_packet([(r, g["mini"][g["mini_off"][r]:g["mini_off"][r + 1]], int(g["bid"][r]), int(g["qlen"][r]))],
                                               par.max_dist_x, par.max_dist_y, pkt_type=fpga.PKT_MINIMIZERS))
            pkts.append(fpga.build_task_packet([(1000 + r, anchors[r])], par.max_dist_x, par.max_dist_y))
        for k, pk in enumerate(pkts):
            drv.submit(pk, tid=k % 4)
        results = drv.wait_results(len(pkts))
    seen = {}
    for raw in results:
        for read_id, err, seeds, mini_pos, rep_len in fpga.parse_result_packet_full(raw):
            assert err == 0
            seen[read_id] = (seeds, mini_pos, rep_len)
    for r in range(n):
        f, p, v, _ = ol.oracle_fpv(par, anchors[r])
        exp = ol.oracle_compact(par, anchors[r], f, p, v)
        assert seen[r][0].tobytes() == exp.tobytes() and seen[1000 + r][0].tobytes() == exp.tobytes(), r
        assert np.array_equal(seen[r][1], g["mini_pos"][g["mp_off"][r]:g["mp_off"][r + 1]]) and seen[r][2] == int(g["rep_len"][r])
        assert len(seen[1000 + r][1]) == 0 and seen[1000 + r][2] == 0


def test_read_beyond_batch_capacity_gets_err_flag_not_exit():
    """An anchor packet whose read is larger than a device batch is answered the reference's way (err_flag = 1, header only,
    map.c:933-944 recomputes it on the host); the other reads of the same stream come back chained."""
    par = P.preset("ava-ont")
    off, a = ag.generate("ava-ont", n_reads=12, seed=77)
    sizes = np.diff(off)
    big = int(np.argmax(sizes))
    cap = int(np.sort(sizes)[-2]) + 1                                   # every read fits but the largest
    with fpga.Driver(bw=par.bw, is_cdna=par.is_cdna, max_skip=par.max_skip, min_sc=par.min_sc, max_anchors_per_batch=cap) as drv:
        for r in range(12):
            drv.submit(fpga.build_task_packet([(r, a[off[r]:off[r + 1]])], par.max_dist_x, par.max_dist_y), tid=r % 3)
        results = drv.wait_results(12)
        st = drv.stats()
    got = {}
    for raw in results:
        for read_id, err, seeds in fpga.parse_result_packet(raw):
            got[read_id] = (err, seeds)
    assert st["err_reads"] == 1 and got[big][0] == 1 and got[big][1] is None
    for r in range(12):
        if r == big:
            continue
        ar = np.ascontiguousarray(a[off[r]:off[r + 1]])
        f, p, v, _ = ol.oracle_fpv(par, ar)
        assert got[r][0] == 0 and got[r][1].tobytes() == ol.oracle_compact(par, ar, f, p, v).tobytes(), r


def test_minimizer_batch_over_capacity_is_split_and_retried():
    """Far more seeds than the admission guess (four per minimizer) and than one device batch holds: the batch is halved until
    its parts fit, only reads that do not fit alone come back with err_flag = 1, every other read is chained as usual."""
    path = [p for p in SEED_FIXTURES if "syn_repeats_mapont" in p][0]
    g = np.load(path, allow_pickle=False)
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    n = len(g["bid"])
    n_anchors = np.diff(g["a_off"])
    cap = int(np.sort(n_anchors)[-2]) + 1                               # the largest read does not fit on its own; no two large ones together
    reads = [(r, g["mini"][g["mini_off"][r]:g["mini_off"][r + 1]], int(g["bid"][r]), int(g["qlen"][r])) for r in range(n)]
    with fpga.Driver(bw=par.bw, is_cdna=par.is_cdna, max_skip=par.max_skip, min_sc=par.min_sc, flag=int(g["flag"]),
                     max_occ=int(g["mid_occ"]), index=[g["img_B"], g["img_H"], g["img_V"], g["img_P"]], max_anchors_per_batch=cap) as drv:
        drv.submit(fpga.build_task_packet(reads, par.max_dist_x, par.max_dist_y, pkt_type=fpga.PKT_MINIMIZERS))   # all reads in ONE packet
        results = drv.wait_results(1)
        st = drv.stats()
    big = int(np.argmax(n_anchors))
    seen = {}
    for raw in results:
        for read_id, err, seeds, mini_pos, rep_len in fpga.parse_result_packet_full(raw):
            seen[read_id] = (err, seeds, mini_pos, rep_len)
    assert sorted(seen) == list(range(n))
    assert seen[big][0] == 1 and st["err_reads"] == int(np.sum(n_anchors > cap)) and st["batches"] > 1
    for r in range(n):
        if n_anchors[r] > cap:
            assert seen[r][0] == 1
            continue
        a = np.ascontiguousarray(g["anchors"][g["a_off"][r]:g["a_off"][r + 1]])
        f, p, v, _ = ol.oracle_fpv(par, a)
        assert seen[r][0] == 0 and seen[r][1].tobytes() == ol.oracle_compact(par, a, f, p, v).tobytes(), r
        assert np.array_equal(seen[r][2], g["mini_pos"][g["mp_off"][r]:g["mp_off"][r + 1]]) and seen[r][3] == int(g["rep_len"][r])


def test_driver_refuses_malformed_submits_and_double_release():
    import ctypes as C
    par = P.preset("ava-ont")
    off, a = ag.generate("ava-ont", n_reads=2, seed=9)
    with fpga.Driver(bw=par.bw, is_cdna=0, max_skip=par.max_skip, min_sc=par.min_sc) as drv:
        L = drv.L
        pkt = fpga.build_task_packet([(0, a[off[0]:off[1]])], par.max_dist_x, par.max_dist_y)
        buf = L.fpga_get_writebuf_thread(len(pkt), 0, 0)
        C.memmove(buf, pkt, len(pkt))
        assert L.fpga_writebuf_submit(buf, len(pkt) + 64, 1) == -1      # more than the buffer was asked for
        assert L.fpga_writebuf_submit(buf, 64 + 32, 1) == -1            # the task header does not fit in what was filled in
        assert L.fpga_writebuf_submit(buf + 8, len(pkt), 1) == -1       # not a driver buffer
        assert L.fpga_writebuf_submit(buf, len(pkt), 1) == 0            # and the packet itself is fine
        (raw,) = drv.wait_results(1)
        assert fpga.parse_result_packet(raw)[0][1] == 0
        # a result buffer can be released once
        n = C.c_int(0)
        drv.submit(pkt)
        res = drv.wait_results(2)
        assert len(res) == 2
        assert L.fpga_release_retbuf(buf) == -1                         # the write buffer went back to the pool with its batch
