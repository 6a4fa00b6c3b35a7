"""chaindp_map_batch on the large dump repeated to ~23 M anchors: a few calls from one context, then from three contexts at once
(what bench.py's map_batch entry times), with the host-side seconds of each phase.  For a kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/map_probe.py [contexts=3] [calls_each=4] [pinned=0]"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minimap2_chaindp_amd import chaindp, params  # noqa: E402

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_each = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pinned = int(sys.argv[3]) if len(sys.argv) > 3 else 0
big = os.path.join(ROOT, "tests", "golden", "_big", "big_avaont.npz")
path = big if os.path.exists(big) else os.path.join(ROOT, "tests", "golden", "seeds", "syn_repeats_avaont.npz")
g = np.load(path, allow_pickle=False)
pv = [int(x) for x in g["params"]]
par = params.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
mult = max(1, int(24_000_000 // max(len(g["anchors"]), 1)))
mini_off = np.concatenate([[0], np.cumsum(np.tile(np.diff(g["mini_off"]), mult))]).astype(np.int64)
mini, bid, qlen = np.tile(g["mini"], (mult, 1)), np.tile(g["bid"], mult), np.tile(g["qlen"], mult)
n_reads = len(bid)
if pinned:                                                       # the caller keeps its minimizers in pinned memory (chaindp_host_alloc)
    keep = []
    def pin(x):
        pa = chaindp.PinnedArray(x.shape, x.dtype); pa.array[...] = x; keep.append(pa); return pa.array
    mini, bid, qlen, mini_off = pin(np.ascontiguousarray(mini, np.uint64)), pin(bid.astype(np.uint32)), pin(qlen.astype(np.int32)), pin(mini_off)
hash_ = (np.arange(n_reads, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(1 << 32)).astype(np.uint32)
cap_a = len(g["anchors"]) * mult + 1024
out = {"input": os.path.relpath(path, ROOT), "reads": n_reads, "minimizers": int(mini_off[-1]), "pinned_input": bool(pinned)}
with chaindp.Device(0, max_anchors=cap_a, max_reads=n_reads + 1) as d:
    ix = d.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
    call = lambda dd: dd.map_batch(ix, int(g["flag"]), int(g["mid_occ"]), par, pv[7], mini_off, mini, bid, qlen, hash_, regs_cap=cap_a // 8)  # noqa: E731
    roff, regs, rep, na = call(d)
    out["anchors"] = int(na)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); call(d); ts.append(time.perf_counter() - t0)
    out["one_context_s"] = ts
    devs = [chaindp.Device(0, max_anchors=cap_a, max_reads=n_reads + 1) for _ in range(n_ctx)]
    try:
        for dd in devs:
            call(dd)
        th = [threading.Thread(target=lambda dd=dd: [call(dd) for _ in range(n_each)]) for dd in devs]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        out["contexts"] = {"n": n_ctx, "batches": n_ctx * n_each, "seconds": dt, "anchors_per_s": na * n_ctx * n_each / dt}
    finally:
        for dd in devs:
            dd.close()
print(json.dumps(out))
