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
with chaindp.Device(0, max_anchors=1 << 23, max_reads=1 << 12) as d:
    ix = d.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    for _ in range(reps):
        t0 = time.time()
        off, a, rep_len, mpo, mp = d.collect_seeds(ix, int(g["flag"]), int(g["mid_occ"]), g["mini_off"], g["mini"], g["bid"], g["qlen"])
        dt = time.time() - t0
    ok = np.array_equal(off, g["a_off"]) and np.array_equal(a, g["anchors"]) and np.array_equal(rep_len, g["rep_len"]) and np.array_equal(mp, g["mini_pos"])
    n = np.diff(off)
    print(f"{len(a)} anchors, {len(n)} reads (max {n.max()}), {dt * 1e3:.1f} ms with transfers, identical to the reference: {ok}")
    sys.exit(0 if ok else 1)
