"""Per-call timing of the streaming pipe on the bench batch (diagnosis of the H2D / D2H overlap)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from minimap2_chaindp_amd import anchorgen, chaindp, params
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
par = params.preset("ava-ont")
off, a = anchorgen.generate("ava-ont", n_reads=n_reads, seed=20261004, threads=16)
total = int(off[-1])
po = chaindp.PinnedArray((n_reads + 1,), np.int64); pa = chaindp.PinnedArray((total, 2), np.uint64)
po.array[:] = off; pa.array[:] = a.reshape(-1, 2)
ev = []
with chaindp.Pipe(0, depth=depth, max_anchors=total + 1, max_reads=n_reads + 1) as pipe:
    nb, sub, done = 10, 0, 0
    t00 = time.perf_counter()
    while done < nb:
        while sub < nb:
            t0 = time.perf_counter()
            ok = pipe.submit(par, po.array, pa.array, tag=sub)
            t1 = time.perf_counter()
            if not ok: break
            ev.append(("submit", sub, round((t0 - t00) * 1e3, 2), round((t1 - t0) * 1e3, 2)))
            sub += 1
        t0 = time.perf_counter()
        tag, soff, seeds = pipe.wait(copy=False)
        t1 = time.perf_counter()
        pipe.release()
        ev.append(("wait", tag, round((t0 - t00) * 1e3, 2), round((t1 - t0) * 1e3, 2)))
        done += 1
    tot = time.perf_counter() - t00
print(json.dumps({"anchors": total, "total_s": tot, "rate": nb * total / tot, "events": ev}))
