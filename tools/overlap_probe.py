#!/usr/bin/env python3
"""Do the bandwidth-bound bookends (prepass, compaction) of one part of a job overlap with the issue-bound chain DP of another
when the parts run on separate contexts (streams)?  The job (ava-ont, --reads) is cut into K parts by anchor count; K host threads
step their part concurrently; aggregate anchors/s against K = 1.   python tools/overlap_probe.py [reads=100000] [steps=10]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from minimap2_chaindp_amd import anchorgen, chaindp, params, shard  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
par = params.preset("ava-ont")
off, a = anchorgen.generate("ava-ont", n_reads=reads, seed=20261004, threads=16)
total = int(off[-1])
out = {"reads": reads, "anchors": total}
for K in (1, 2, 3, 4):
    cuts = shard.split_by_anchors(off, K)
    devs = []
    for k in range(K):
        o = np.ascontiguousarray(off[cuts[k]:cuts[k + 1] + 1] - off[cuts[k]])
        aa = a[int(off[cuts[k]]):int(off[cuts[k + 1]])]
        d = chaindp.Device(0, max_anchors=int(o[-1]) + 1, max_reads=len(o))
        d.upload(o, aa)
        for _ in range(2):
            d.run_full(par)
        d.sync()
        devs.append(d)

    def work(d):
        for _ in range(steps):
            d.run_full(par)
        d.sync()
    th = [threading.Thread(target=work, args=(d,)) for d in devs]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    out[f"contexts_{K}"] = {"ms_per_job_step": dt / steps * 1e3, "anchors_per_s": total * steps / dt}
    for d in devs:
        d.close()
print(json.dumps(out))
