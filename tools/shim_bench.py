#!/usr/bin/env python3
"""Throughput of the packet path (reference fpga.h ABI served by the GPU), PCIe-inclusive: anchor packets
of 8 reads from P producer threads, result packets drained by one receiver.  Not the headline metric
(bench.py's value is HBM-resident); recorded in DESIGN.md section 6."""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimap2_chaindp_amd import anchorgen as ag, fpga, params as P  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
par = P.preset("ava-ont")
off, a = ag.generate("ava-ont", n_reads=n_reads, seed=3, threads=16)
packets = [fpga.build_task_packet([(r, a[off[r]:off[r + 1]]) for r in range(k, min(k + 8, n_reads))],
                                  par.max_dist_x, par.max_dist_y) for k in range(0, n_reads, 8)]
tot = int(off[-1])
with fpga.Driver(bw=par.bw, is_cdna=0, max_skip=par.max_skip, min_sc=par.min_sc, max_packets_per_batch=256) as drv:
    for rep in range(3):
        drv.results.clear()
        t0 = time.time()

        def producer(tid):
            for k in range(tid, len(packets), n_threads):
                drv.submit(packets[k], tid)
        ths = [threading.Thread(target=producer, args=(t,)) for t in range(n_threads)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        drv.wait_results(len(packets) * (rep + 1) if False else len(packets))
        dt = time.time() - t0
        print(f"rep {rep}: {n_reads} reads, {tot} anchors, {len(packets)} packets in {dt*1e3:.1f} ms -> "
              f"{tot/dt/1e6:.1f} M anchors/s  stats={drv.stats()}", flush=True)
