"""Seed collection alone (probe, expand, sort) on the large dump repeated to ~23 M anchors, timed over a few calls:
   python tools/seed_sort_probe.py     (CHAINDP_LIB selects the build)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minimap2_chaindp_amd import chaindp  # noqa: E402

big = os.path.join(ROOT, "tests", "golden", "_big", "big_avaont.npz")
path = big if os.path.exists(big) else os.path.join(ROOT, "tests", "golden", "seeds", "syn_repeats_avaont.npz")
g = np.load(path, allow_pickle=False)
mult = max(1, int(24_000_000 // max(len(g["anchors"]), 1)))
mini_off = np.concatenate([[0], np.cumsum(np.tile(np.diff(g["mini_off"]), mult))]).astype(np.int64)
mini, bid, qlen = np.tile(g["mini"], (mult, 1)), np.tile(g["bid"], mult), np.tile(g["qlen"], mult)
n_reads = len(bid)
cap_a = len(g["anchors"]) * mult + 1024
with chaindp.Device(0, max_anchors=cap_a, max_reads=n_reads + 1) as d:
    ix = d.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        d.collect_seeds(ix, int(g["flag"]), int(g["mid_occ"]), mini_off, mini, bid, qlen)
        d.sync()
        ts.append(time.perf_counter() - t0)
print(json.dumps({"lib": os.environ.get("CHAINDP_LIB", "main"), "reads": n_reads, "seconds": ts}))
