#!/usr/bin/env python3
"""Generates tests/golden/seeds/*.npz: what the reference hands to its device for the seed-collection step and what
its own collect_seed_hits (map.c:187-236) returns for it.  Produced in the build container by oracle/_ref/mt_dump
(`make -C oracle ref-dump`), i.e. by the UNMODIFIED reference's sketch -> index (incl. the FPGA index image of
index.c:603-720) -> collect_seed_hits; the GPU box and the CPU test tier only read the committed .npz files.

Inputs: the reference's own test/*.fa, plus a seeded synthetic genome with dispersed, inverted and tandem repeats and
12 noisy reads of it (written to /tmp by this script), mapped read-to-genome and all-vs-all, so that multi-occurrence
minimizers (the P array), equal-x ties in the sort, tandem / self flags, NO_DIAG / NO_DUAL skipping and a non-zero
rep_len all occur.

Each fixture: flag, mid_occ, img_B/H/V/P (uint8), params[int32 8] (max_dist_x, max_dist_y, bw, max_skip, min_sc,
is_cdna, n_segs, min_cnt), bid[R], qlen[R], mini_off[R+1] + mini[uint64 *,2], a_off[R+1] + anchors[uint64 *,2],
rep_len[R], mp_off[R+1] + mini_pos[uint64 *]."""
import os
import random
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
DUMP = os.path.join(ROOT, "oracle", "_ref", "mt_dump")
OUT = os.path.join(HERE, "seeds")


def read_seed_dump(path):
    b = open(path, "rb").read()
    assert b[:8] == b"SEEDDMP1"
    flag, mid_occ, n_reads = struct.unpack_from("<iii", b, 8)
    o, img = 20, []
    for _ in range(4):
        (nb,) = struct.unpack_from("<q", b, o); o += 8
        img.append(np.frombuffer(b, np.uint8, nb, o).copy()); o += nb
    reads = []
    for _ in range(n_reads):
        bid, qlen = struct.unpack_from("<Ii", b, o); o += 8
        (nm,) = struct.unpack_from("<q", b, o); o += 8
        mv = np.frombuffer(b, np.uint64, nm * 2, o).reshape(-1, 2).copy(); o += nm * 16
        (na,) = struct.unpack_from("<q", b, o); o += 8
        a = np.frombuffer(b, np.uint64, na * 2, o).reshape(-1, 2).copy(); o += na * 16
        rep_len, nmp = struct.unpack_from("<ii", b, o); o += 8
        mp = np.frombuffer(b, np.uint64, nmp, o).copy(); o += nmp * 8
        reads.append((bid, qlen, mv, a, rep_len, mp))
    assert o == len(b)
    return flag, mid_occ, img, reads


def read_params(path):
    """DP arguments of the first read in the companion ANCHDMP1 file (they are per preset, not per read)."""
    raw = open(path, "rb").read()
    (n_reads,) = struct.unpack_from("<i", raw, 8)
    return np.array(struct.unpack_from("<8i", raw, 12), np.int32) if n_reads else np.zeros(8, np.int32)


def cat(arrs, width=None):
    off = np.zeros(len(arrs) + 1, np.int64)
    for k, x in enumerate(arrs):
        off[k + 1] = off[k] + len(x)
    if width:
        data = np.concatenate(arrs) if off[-1] else np.zeros((0, width), np.uint64)
    else:
        data = np.concatenate(arrs) if off[-1] else np.zeros(0, np.uint64)
    return off, data


def make(name, preset, target, query):
    a_path, s_path = f"/tmp/seedgold_{name}.dump", f"/tmp/seedgold_{name}.seed"
    subprocess.run([DUMP, preset, target, query, a_path, s_path], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    flag, mid_occ, img, reads = read_seed_dump(s_path)
    mini_off, mini = cat([r[2] for r in reads], 2)
    a_off, anchors = cat([r[3] for r in reads], 2)
    mp_off, mini_pos = cat([r[5] for r in reads])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), flag=np.int32(flag), mid_occ=np.int32(mid_occ),
                        img_B=img[0], img_H=img[1], img_V=img[2], img_P=img[3], params=read_params(a_path),
                        bid=np.array([r[0] for r in reads], np.uint32), qlen=np.array([r[1] for r in reads], np.int32),
                        mini_off=mini_off, mini=mini, a_off=a_off, anchors=anchors,
                        rep_len=np.array([r[4] for r in reads], np.int32), mp_off=mp_off, mini_pos=mini_pos)
    ties = sum(int((np.diff(r[3][:, 0]) == 0).sum()) for r in reads if len(r[3]) > 1)
    print(f"{name}: flag={flag:#x} mid_occ={mid_occ} reads={len(reads)} minimizers={len(mini)} anchors={len(anchors)} "
          f"P={len(img[3]) // 8} x-ties={ties} rep_len={[r[4] for r in reads][:6]}")


def synthetic_fasta():
    rnd = random.Random(11)

    def seq(n):
        return "".join(rnd.choice("ACGT") for _ in range(n))

    def mutate(s, rate):
        out = []
        for c in s:
            r = rnd.random()
            if r < rate / 3:
                continue
            if r < 2 * rate / 3:
                out.append(rnd.choice("ACGT"))
            elif r < rate:
                out.append(c); out.append(rnd.choice("ACGT"))
            else:
                out.append(c)
        return "".join(out)

    def rc(s):
        return s[::-1].translate(str.maketrans("ACGT", "TGCA"))

    unit, tand = seq(400), seq(37)
    g = seq(6000) + unit + seq(5000) + tand * 12 + seq(4000) + unit + seq(3000) + rc(unit) + seq(6000) + unit + seq(2000)
    open("/tmp/seedgold_tgt.fa", "w").write(">chrS\n" + g + "\n")
    reads = []
    for i in range(12):
        st, ln = rnd.randrange(0, len(g) - 3500), rnd.randrange(2000, 3500)
        s = mutate(g[st:st + ln], 0.06)
        reads.append(("r%02d" % i, rc(s) if i % 3 == 0 else s))
    open("/tmp/seedgold_reads.fa", "w").write("".join(">%s\n%s\n" % x for x in reads))
    return "/tmp/seedgold_tgt.fa", "/tmp/seedgold_reads.fa"


def big_fasta():
    """600 noisy 8 kb reads of a 150 kb random genome (coverage 32), both strands: the all-vs-all batch of
    tests/golden/_big/big_avaont.npz (42 MB, git-ignored; travels to the GPU box with the snapshot)."""
    rnd = random.Random(5)
    g = "".join(rnd.choice("ACGT") for _ in range(150000))

    def mutate(s, rate):
        out = []
        for c in s:
            r = rnd.random()
            if r < rate / 3:
                continue
            if r < 2 * rate / 3:
                out.append(rnd.choice("ACGT"))
            elif r < rate:
                out.append(c); out.append(rnd.choice("ACGT"))
            else:
                out.append(c)
        return "".join(out)

    reads = []
    for i in range(600):
        st = rnd.randrange(0, len(g) - 8000)
        s = mutate(g[st:st + 8000], 0.08)
        reads.append(("r%04d" % i, s[::-1].translate(str.maketrans("ACGT", "TGCA")) if i % 2 else s))
    open("/tmp/seedgold_big.fa", "w").write("".join(">%s\n%s\n" % x for x in reads))
    return "/tmp/seedgold_big.fa"


if __name__ == "__main__":
    import sys
    if "--big" in sys.argv:                      # the large, git-ignored dump for the at-scale test and tools/shim_minimizer_bench.py
        OUT = os.path.join(HERE, "_big")
        os.makedirs(OUT, exist_ok=True)
        fa = big_fasta()
        make("big_avaont", "ava-ont", fa, fa)
        sys.exit(0)
    os.makedirs(OUT, exist_ok=True)
    t = os.path.join(REF, "test")
    make("mt_orang_vs_human_mapont", "map-ont", f"{t}/MT-human.fa", f"{t}/MT-orang.fa")
    make("mt_human_vs_orang_avaont", "ava-ont", f"{t}/MT-orang.fa", f"{t}/MT-human.fa")
    make("mt_human_self_avaont", "ava-ont", f"{t}/MT-human.fa", f"{t}/MT-human.fa")
    make("inv_mapont", "map-ont", f"{t}/t-inv.fa", f"{t}/q-inv.fa")
    make("inv_sr", "sr", f"{t}/t-inv.fa", f"{t}/q-inv.fa")
    tgt, reads = synthetic_fasta()
    make("syn_repeats_mapont", "map-ont", tgt, reads)
    make("syn_repeats_avaont", "ava-ont", reads, reads)
    make("syn_repeats_avapb", "ava-pb", reads, reads)
