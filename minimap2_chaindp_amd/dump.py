"""Anchor-dump files: what the reference's dump branch would hand to the accelerator.

The reference names a BRANCH_MINIMAP2_DUMP_CHAINDP (its README.md:10) that is not in the tree.  oracle/mt_dump.c
plays that role in the build container: it drives the reference's own sketch -> index -> collect_seed_hits
(map.c:87-236) on a target/query FASTA pair and writes, per query read, the sorted mm128_t anchors that
mm_chain_dp_fpga (chain.c:218) receives, with that read's DP arguments.  This module reads such files
 so that real-data batches can be chained on the GPU and checked like the synthetic ones.

Format (little endian): magic "ANCHDMP1", int32 n_reads, then per read
  int32 max_dist_x, max_dist_y, bw, max_skip, min_sc, is_cdna, n_segs, min_cnt;  int64 n;  n x (uint64 x, uint64 y).
"""
import struct

import numpy as np

from .params import ChainParams

MAGIC = b"ANCHDMP1"


def read_dump(path):
    """-> list of (ChainParams, min_cnt, anchors uint64[n,2]) per read."""
    raw = open(path, "rb").read()
    if raw[:8] != MAGIC:
        raise ValueError(f"{path}: not an anchor dump")
    (n_reads,) = struct.unpack_from("<i", raw, 8)
    pos, out = 12, []
    for _ in range(n_reads):
        hdr = struct.unpack_from("<8i", raw, pos)
        (n,) = struct.unpack_from("<q", raw, pos + 32)
        pos += 40
        a = np.frombuffer(raw, np.uint64, n * 2, pos).reshape(n, 2).copy()
        pos += n * 16
        out.append((ChainParams(*hdr[:7]), hdr[7], a))
    return out


def write_dump(path, reads):
    """reads: iterable of (ChainParams, min_cnt, anchors uint64[n,2])."""
    reads = list(reads)
    with open(path, "wb") as fh:
        fh.write(MAGIC + struct.pack("<i", len(reads)))
        for par, min_cnt, a in reads:
            a = np.ascontiguousarray(a, np.uint64).reshape(-1, 2)
            fh.write(struct.pack("<8i", *par.astuple(), min_cnt) + struct.pack("<q", a.shape[0]) + a.tobytes())


def batches(reads):
    """Groups a dump's reads by identical DP arguments -> list of (ChainParams, min_cnt, off, anchors, read indices)."""
    groups = {}
    for k, (par, min_cnt, a) in enumerate(reads):
        groups.setdefault((par.astuple(), min_cnt), []).append((k, a))
    out = []
    for (pt, min_cnt), items in groups.items():
        off = np.zeros(len(items) + 1, np.int64)
        off[1:] = np.cumsum([a.shape[0] for _, a in items])
        anchors = np.concatenate([a for _, a in items]) if off[-1] else np.zeros((0, 2), np.uint64)
        out.append((ChainParams(*pt), min_cnt, off, anchors, [k for k, _ in items]))
    return out
