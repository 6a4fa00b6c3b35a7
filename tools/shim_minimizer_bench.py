#!/usr/bin/env python3
"""Throughput of the reference's own packet type (minimizers in, new_seed[] + mini_pos[] out) through the driver ABI:
index image loaded with fpga_load_index, seeds collected and chained on the GPU.  Input: a seed dump made in the build
container by tests/golden/make_seed_golden.py-style tooling (tests/golden/_big/big_avaont.npz, git-ignored).  PCIe-inclusive;
recorded in DESIGN.md, never the headline metric."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimap2_chaindp_amd import fpga, params as P  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "_big", "big_avaont.npz")
reps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8          # the batch is replayed `reps` times back to back
n_threads = 8
g = np.load(path, allow_pickle=False)
par = P.preset("ava-ont")
n_reads = len(g["bid"])
reads = [(r, g["mini"][g["mini_off"][r]:g["mini_off"][r + 1]], int(g["bid"][r]), int(g["qlen"][r])) for r in range(n_reads)]
packets = [fpga.build_task_packet(reads[k:k + 8], par.max_dist_x, par.max_dist_y, pkt_type=fpga.PKT_MINIMIZERS) for k in range(0, n_reads, 8)]
tot_a, tot_m = int(g["a_off"][-1]), int(g["mini_off"][-1])
if "--write-replay" in sys.argv:          # the same packets and index image as a file for tools/shim_replay.c (no Python in the loop)
    import struct
    out = sys.argv[sys.argv.index("--write-replay") + 1]
    with open(out, "wb") as fh:
        fh.write(b"SHIMRPL1" + struct.pack("<6i", int(g["flag"]), int(g["mid_occ"]), par.bw, par.max_skip, par.min_sc, len(packets)))
        for k in ("img_B", "img_H", "img_V", "img_P"):
            blob = np.ascontiguousarray(g[k], np.uint8)
            fh.write(struct.pack("<q", blob.size)); fh.write(blob.tobytes())
        for pk in packets:
            fh.write(struct.pack("<I", len(pk))); fh.write(pk)
    print(f"wrote {out}: {len(packets)} packets, {tot_m} minimizers, {tot_a} anchors expected")
    sys.exit(0)
with fpga.Driver(bw=par.bw, is_cdna=0, max_skip=par.max_skip, min_sc=par.min_sc, flag=int(g["flag"]), max_occ=int(g["mid_occ"]),
                 index=[g["img_B"], g["img_H"], g["img_V"], g["img_P"]], max_packets_per_batch=256) as drv:
    for rep in range(3):
        drv.results.clear()
        t0 = time.time()

        def producer(tid):
            for _ in range(reps):
                for k in range(tid, len(packets), n_threads):
                    drv.submit(packets[k], tid)
        ths = [threading.Thread(target=producer, args=(t,)) for t in range(n_threads)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        drv.wait_results(len(packets) * reps)
        dt = time.time() - t0
        print(f"rep {rep}: {n_reads * reps} reads, {tot_m * reps} minimizers -> {tot_a * reps} anchors, {len(packets) * reps} packets in {dt * 1e3:.1f} ms -> "
              f"{tot_a * reps / dt / 1e6:.1f} M anchors/s ({tot_m * reps / dt / 1e6:.1f} M minimizers/s)  stats={drv.stats()}", flush=True)
