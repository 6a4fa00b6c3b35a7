"""CPU tier: the parts of bench.py that are plain host logic -- the issue-rate ceilings read from the committed calibration runs,
the stored PMC figures (quoted only for the build they were measured on, scaled per anchor to a rank's share at N > 1), the
replay-file writer of the packet-ABI entry, and the strong-scaling cut of the job."""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from minimap2_chaindp_amd import params as P, shard  # noqa: E402


def test_issue_ceilings_come_from_the_committed_calibration():
    c = bench.issue_ceilings()
    assert c is not None and len(c["files"]) >= 1
    assert 850 < c["valu_full"] < 1100 and 520 < c["valu_half"] < 650 and 500 < c["salu"] < 650      # G wave-instructions/s on MI355X
    assert c["valu_half"] < c["valu_full"]


def test_stored_pmc_is_keyed_on_the_kernel_sources_and_scaled_per_anchor(tmp_path, monkeypatch):
    t = bench.stored_pmc_any()
    assert t is not None and "kernel_source_sha16" in t
    n0 = t["anchors_per_launch"]
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda: t["kernel_source_sha16"])
    same = bench.stored_pmc(n0)
    assert same is not None and "scaled_from_anchors" not in same and same["hbm_bytes_per_launch"] == t["hbm_bytes_per_launch"]
    half = bench.stored_pmc(n0 // 2)
    assert half["scaled_from_anchors"] == n0 and abs(half["hbm_bytes_per_launch"] * 2 - t["hbm_bytes_per_launch"]) < 1e-6 * t["hbm_bytes_per_launch"] + 64
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda: "0" * 16)               # another build of the kernels: nothing is quoted
    assert bench.stored_pmc(n0) is None and bench.measured_traffic(n0) is None and bench.measured_issue(n0, 1.0) is None


def test_replay_file_layout(tmp_path):
    par = P.preset("ava-ont")
    packets = [b"\x01" * 64 + b"\x02" * 128, b"\x03" * 64]
    path = str(tmp_path / "x.rpl")
    bench._replay_file(path, packets, [np.arange(5, dtype=np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.uint8)], 7, 9, par)
    raw = open(path, "rb").read()
    assert raw[:8] == b"SHIMRPL1"
    flag, mid_occ, bw, max_skip, min_sc, n = struct.unpack_from("<6i", raw, 8)
    assert (flag, mid_occ, bw, max_skip, min_sc, n) == (7, 9, par.bw, par.max_skip, par.min_sc, 2)
    pos = 32
    sizes = []
    for _ in range(4):
        (nb,) = struct.unpack_from("<q", raw, pos)
        sizes.append(nb)
        pos += 8 + nb
    assert sizes == [5, 0, 0, 0]
    for pk in packets:
        (sz,) = struct.unpack_from("<I", raw, pos)
        assert sz == len(pk) and raw[pos + 4:pos + 4 + sz] == pk
        pos += 4 + sz
    assert pos == len(raw)


def test_job_cuts_tile_the_job_and_balance_anchors():
    cuts = shard.job_cuts("ava-ont", 4, 64, seed=3, threads=2)
    assert cuts[0] == 0 and cuts[-1] == 64 and np.all(np.diff(cuts) > 0)
    from minimap2_chaindp_amd import anchorgen
    off = anchorgen.offsets("ava-ont", n_reads=64, seed=3, threads=2)
    loads = np.diff(off[cuts])
    assert loads.max() - loads.min() <= 2 * int(np.diff(off).max())
    o, a, first = shard.generate_job_shard("ava-ont", 1, 4, 64, 3, threads=2)
    assert first == int(cuts[1]) and len(o) - 1 == int(cuts[2] - cuts[1]) and int(o[-1]) == int(off[cuts[2]] - off[cuts[1]]) == len(a)
