#!/usr/bin/env python3
"""chaindp_map_batch on the reference's all-vs-all dump (tests/golden/_big, repeated to ~23 M anchors), a few calls: for
rocprofv3 --kernel-trace --stats (which kernels the minimizers-in / hits-out call spends its time in).
  python tools/map_batch_probe.py [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from minimap2_chaindp_amd import chaindp, params  # noqa: E402

big = os.path.join(ROOT, "tests", "golden", "_big", "big_avaont.npz")
path = big if os.path.exists(big) else os.path.join(ROOT, "tests", "golden", "seeds", "syn_repeats_avaont.npz")
g = np.load(path, allow_pickle=False)
pv = [int(x) for x in g["params"]]
par = params.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
mult = max(1, int(24_000_000 // max(len(g["anchors"]), 1)))
mini_off = np.concatenate([[0], np.cumsum(np.tile(np.diff(g["mini_off"]), mult))]).astype(np.int64)
mini, bid, qlen = np.tile(g["mini"], (mult, 1)), np.tile(g["bid"], mult), np.tile(g["qlen"], mult)
n_reads = len(bid)
hash_ = (np.arange(n_reads, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(1 << 32)).astype(np.uint32)
cap_a = len(g["anchors"]) * mult + 1024
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
with chaindp.Device(0, max_anchors=cap_a, max_reads=n_reads + 1) as d:
    ix = d.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    d.map_batch(ix, int(g["flag"]), int(g["mid_occ"]), par, pv[7], mini_off, mini, bid, qlen, hash_, regs_cap=cap_a // 8)
    t0 = time.perf_counter()
    for _ in range(n):
        roff, regs, rep, na = d.map_batch(ix, int(g["flag"]), int(g["mid_occ"]), par, pv[7], mini_off, mini, bid, qlen, hash_, regs_cap=cap_a // 8)
    dt = (time.perf_counter() - t0) / n
print(f"{os.path.basename(path)} x{mult}: {n_reads} reads, {int(mini_off[-1])} minimizers, {na} anchors, {int(roff[-1])} hits; {dt * 1e3:.2f} ms per call, {na / dt / 1e9:.3f} G anchors/s")
