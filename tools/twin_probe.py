"""Times the chain DP kernel variants on one batch and reports how many units the two-per-wave kernel hands over."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from minimap2_chaindp_amd import anchorgen, chaindp, params

preset = sys.argv[1] if len(sys.argv) > 1 else "ava-ont"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
par = params.preset(preset if preset in params.PRESETS else "ava-ont")
off, a = anchorgen.generate(preset, n_reads=n_reads, seed=20261004, threads=16)
total = int(off[-1])
out = {"preset": preset, "reads": n_reads, "anchors": total}
with chaindp.Device(0, max_anchors=total + 1, max_reads=n_reads + 1) as dev:
    dev.upload(off, a)
    ref = None
    for variant in (2, 0):
        dev.set_variant(variant)
        for _ in range(3):
            dev.run_full(par)
        dev.sync()
        dev.set_profiling(True); dev.kernel_ms(reset=True)
        for _ in range(10):
            dev.run_full(par)
        dev.sync()
        k = dev.kernel_ms(reset=True)
        dev.set_profiling(False)
        st = dev.stats()
        f, p, v = dev.download()
        if ref is None:
            ref = (f, p, v)
        same = all(np.array_equal(x, y) for x, y in zip(ref, (f, p, v)))
        out[f"variant{variant}"] = {"chain_dp_ms": k["chain_dp"][0] / k["chain_dp"][1], "prepass_ms": k["prepass"][0] / k["prepass"][1],
                                    "compact_ms": k["compact"][0] / k["compact"][1], "units": st["units"],
                                    "leftover": dev.leftover_units() if variant == 0 else None, "same_as_variant2": same}
print(json.dumps(out))
