"""Seed collection on the large reference dump (tests/golden/_big, made by tests/golden/make_seed_golden.py --big), a few
times over, for rocprofv3: `rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/seed_scale_run.py`.
Checks the result against the dump each time it runs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minimap2_chaindp_amd import chaindp  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "_big", "big_avaont.npz"), allow_pickle=False)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mult = int(sys.argv[2]) if len(sys.argv) > 2 else 1              # the batch repeated `mult` times over (a larger batch of the same reads)
mini_off = np.concatenate([[0], np.cumsum(np.tile(np.diff(g["mini_off"]), mult))]).astype(np.int64)
mini, bid, qlen = np.tile(g["mini"], (mult, 1)), np.tile(g["bid"], mult), np.tile(g["qlen"], mult)
with chaindp.Device(0, max_anchors=(1 << 23) * mult, max_reads=(1 << 12) * mult) as d:
    ix = d.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    for _ in range(reps):
        t0 = time.time()
        off, a, rep_len, mpo, mp = d.collect_seeds(ix, int(g["flag"]), int(g["mid_occ"]), mini_off, mini, bid, qlen)
        dt = time.time() - t0
    n1 = len(g["anchors"])
    ok = (np.array_equal(np.diff(off), np.tile(np.diff(g["a_off"]), mult)) and all(np.array_equal(a[k * n1:(k + 1) * n1], g["anchors"]) for k in range(mult))
          and np.array_equal(rep_len, np.tile(g["rep_len"], mult)) and np.array_equal(mp, np.tile(g["mini_pos"], mult)))
    n = np.diff(off)
    print(f"{len(a)} anchors, {len(n)} reads (max {n.max()}), {dt * 1e3:.1f} ms with transfers, identical to the reference: {ok}")
    sys.exit(0 if ok else 1)
