#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory: per-kernel average duration (kernel trace) and
per-kernel average counter values (PMC passes), plus the derived per-anchor figures for the chain
DP kernel.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (it reports half of
the bytes of wide coalesced reads); FETCH_SIZE/WRITE_SIZE are in KiB."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha16():
    """Hash of the chain DP kernels' sources: bench.py only quotes stored PMC figures for the build they were measured on."""
    h = hashlib.sha256()
    for f in ("chaindp_twin.hip", "chaindp_kernels.hip", "chaindp_fast.h", "chaindp_wave.h", "chaindp_lanes.h"):
        h.update(open(os.path.join(ROOT, "minimap2_chaindp_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


short = lambda n: n.split("(")[0].replace("void ", "").replace("chaindp::", "")
print(f"# profile summary of {os.path.basename(d)}")
try:
    b = json.loads(open(os.path.join(d, "bench_under_trace.json")).read().strip().splitlines()[-1])
    anchors = b["config"].get("anchors_on_rank0") or b["config"]["anchors_per_gpu"]
    print(f"bench under trace: value={b['value']:.4g} anchors/s, kernel_ms={b['kernel_ms']}, anchors/launch={anchors}")
except Exception as e:  # noqa: BLE001
    anchors = None
    print("no bench json:", e)
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("\n## kernel trace (rocprofv3 --kernel-trace --stats)")
    print(f"{'kernel':40s} {'calls':>6s} {'avg_us':>10s} {'pct':>7s}")
    for r in csv.DictReader(open(f)):
        print(f"{short(r['Name'])[:40]:40s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):7.2f}")
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\n## PMC (average per launch)")
for k, cs in agg.items():
    if not k.startswith("k_"):
        continue
    print(f"{k}:")
    for c, v in sorted(cs.items()):
        print(f"    {c:26s} {sum(v)/len(v):18.1f}")
for ck in [k for k in agg if k.startswith("k_chain_twin") or k.startswith("k_chain_units")]:
  if anchors:
      c = {k: sum(v) / len(v) for k, v in agg[ck].items()}
      print(f"\n## {ck}: derived (anchors per launch = {anchors})")
      for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"):
          if name in c:
              print(f"    {name}/anchor = {c[name]/anchors:.2f}")
      if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
          rd, wr = 2 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
          print(f"    HBM traffic per launch: read {rd/1e6:.1f} MB (FETCH_SIZE x2 gfx950 correction), write {wr/1e6:.1f} MB,"
                f" total {(rd+wr)/anchors:.1f} B/anchor (algorithmic 24 B/anchor)")
          with open(os.path.join(d, "traffic.json" if ck.startswith("k_chain_twin") or not any(k.startswith("k_chain_twin") for k in agg) else "traffic_units.json"), "w") as fh:
              json.dump({"kernel": ck, "anchors_per_launch": anchors, "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                         "hbm_bytes_per_launch": rd + wr,
                         "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
                         "lds_insts_per_launch": c.get("SQ_INSTS_LDS"), "kernel_source_sha16": kernel_source_sha16(),
                         "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KiB -> bytes, "
                                   "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B), average over launches"}, fh)
      if "GRBM_GUI_ACTIVE" in c:
          print(f"    GRBM_GUI_ACTIVE/8 = {c['GRBM_GUI_ACTIVE']/8/1e6:.2f} M cycles")
