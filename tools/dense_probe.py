#!/usr/bin/env python3
"""Dense-shape probe (GPU box): chain DP time of a batch of dense-repeat reads under the kernel's routing options.
  python tools/dense_probe.py [reads ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minimap2_chaindp_amd import anchorgen as ag, chaindp, params  # noqa: E402

par = params.preset("ava-ont")
for n_reads in [int(x) for x in sys.argv[1:]] or [100]:
    off, a = ag.generate("dense", n_reads=n_reads, seed=20261004)
    total = int(off[-1])
    with chaindp.Device(0, max_anchors=total + 1, max_reads=n_reads + 1) as dev:
        dev.upload(off, a)
        cases = (("k_chain_units alone", 128, False), ("handover, as the batch decides", 128, True), ("handover to k_chain_dense", 128, 2), ("handover to k_chain_dense1", 128, 3), ("handover to k_chain_dense16", 128, 4))
        if os.environ.get("CHAINDP_LIB"):
            cases = cases[1:2]
        for label, ring, handover in cases:
            dev.set_ring(ring); dev.set_deep_handover(handover)
            dev.run_full(par); dev.sync()
            dev.set_profiling(True); dev.kernel_ms(reset=True)
            t0 = time.perf_counter()
            for _ in range(2):
                dev.run_full(par)
            dev.sync()
            dt = (time.perf_counter() - t0) / 2
            kms = dev.kernel_ms(reset=True); dev.set_profiling(False)
            print(f"reads={n_reads} anchors={total} {label:24s} step {dt * 1e3:9.2f} ms  chain_dp {kms['chain_dp'][0] / max(kms['chain_dp'][1], 1):9.2f} ms  "
                  f"{total / dt / 1e9:7.4f} G anchors/s  leftover={dev.leftover_units()} deep_units={dev.deep_units()}", flush=True)
