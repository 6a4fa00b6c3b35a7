"""Times chaindp_backtrack's kernels on the bench shard (12 500 ava-ont reads), a few calls in a row:  python tools/bt_probe.py [reads]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minimap2_chaindp_amd import chaindp, params as P, shard  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
par = P.preset("ava-ont")
off, a = shard.generate_shard("ava-ont", 0, 1, reads, 20261004, threads=8)
tot = int(off[-1])
out = {"reads": reads, "anchors": tot, "calls": []}
with chaindp.Device(0, max_anchors=tot + 1, max_reads=reads + 1) as d:
    d.upload(off, a)
    d.run_full(par)
    d.sync()
    for k in range(4):
        d.set_profiling(True); d.kernel_ms(reset=True)
        t0 = time.perf_counter()
        coff, u, boff, b = d.backtrack(par, 3)
        dt = time.perf_counter() - t0
        kb = d.kernel_ms(reset=True)["backtrack"]
        out["calls"].append({"kernels_ms": kb[0] / max(kb[1], 1), "with_download_s": dt, "chains": int(coff[-1])})
print(json.dumps(out))
