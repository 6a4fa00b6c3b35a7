"""Quick device-side timing used while bringing the kernels up (not a test, not the benchmark)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
from minimap2_chaindp_amd import anchorgen as ag, chaindp, params as P

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
gen = sys.argv[2] if len(sys.argv) > 2 else "ava-ont"
preset = sys.argv[3] if len(sys.argv) > 3 else gen
t0 = time.time()
off, a = ag.generate(gen, n_reads=n_reads, seed=5)
print(f"generated {n_reads} reads, {int(off[-1])} anchors in {time.time()-t0:.1f}s", flush=True)
par = P.preset(preset)
with chaindp.Device(0, max_anchors=int(off[-1]) + 1, max_reads=n_reads + 1) as d:
    d.upload(off, a)
    for ring in (128, 256, 512):
        d._check(d._lib.chaindp_set_ring(d._ctx, ring))
        d.set_profiling(True)
        d.run(par); d.sync(); d.kernel_ms(reset=True)
        for _ in range(5):
            d.run(par)
        d.sync()
        ms = d.kernel_ms(reset=True)
        tot = int(off[-1])
        dp = ms["chain_dp"][0] / ms["chain_dp"][1]; pre = ms["prepass"][0] / ms["prepass"][1]
        print(f"ring {ring}: prepass {pre:.3f} ms, chain_dp {dp:.3f} ms -> {tot/ (dp+pre) / 1e6:.1f} M anchors/s "
              f"({tot*24/(dp+pre)/1e6:.1f} GB/s algorithmic), stats {d.stats()}", flush=True)
