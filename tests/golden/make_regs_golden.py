#!/usr/bin/env python3
"""Generates tests/golden/regs/*.npz: what the UNMODIFIED reference makes of the chains of every seed fixture
(tests/golden/seeds/*.npz) -- mm_chain_dp_fpga + mm_chain_dp_bottom (chain.c), mm_gen_regs (hit.c:52-95) and mm_est_err
(esterr.c:30-64), all called in oracle/_ref/ (`make -C oracle ref`, build container only).  The CPU tier checks the
oracle's restatement against these files, the GPU tier the kernels; neither needs the reference at run time.

Each fixture: min_cnt, hash[R] (the per-read `hash` argument), qlen[R], ref_len[n_ref], chains_off[R+1] + u[uint64],
b_off[R+1] + b[uint64 *,2] (the chains), regs[uint8 *,80] (mm_gen_regs' records, read after read, the 72 bytes of
fields and 8 zero bytes where the mm_extra_t pointer is), regs_div[uint8 *,80] (the same records after mm_est_err with the
fixture's mini_pos)."""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from minimap2_chaindp_amd import params as P  # noqa: E402

OUT = os.path.join(HERE, "regs")
os.makedirs(OUT, exist_ok=True)
assert ol.have_ref(), "build oracle/_ref first (make -C oracle ref)"
for path in sorted(glob.glob(os.path.join(HERE, "seeds", "*.npz"))):
    g = np.load(path, allow_pickle=False)
    pv = [int(x) for x in g["params"]]
    par = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
    R = len(g["qlen"])
    hash_ = (np.arange(R, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(1 << 32)).astype(np.uint32)
    us, bs, regs = [], [], []
    for r in range(R):
        a = np.ascontiguousarray(g["anchors"][g["a_off"][r]:g["a_off"][r + 1]])
        if len(a):
            _, _, _, seeds = ol.ref_fpv_seeds(par, a)
            u, b = ol.ref_bottom(pv[7], par.min_sc, 1, seeds)
        else:
            u, b = np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
        us.append(u); bs.append(b.reshape(-1, 2))
        regs.append(ol.ref_gen_regs(int(hash_[r]), int(g["qlen"][r]), u, b))
    allr = np.concatenate(regs) if regs else np.zeros(0, ol.REG_DTYPE)
    n_ref = int(allr["rid"].max()) + 1 if len(allr) else 1
    ref_len = (np.arange(n_ref, dtype=np.int64) * 37 % 5000 + (int(allr["re"].max()) if len(allr) else 0) - 2000).astype(np.int32)
    divs = []
    for r in range(R):
        mp = np.ascontiguousarray(g["mini_pos"][g["mp_off"][r]:g["mp_off"][r + 1]])
        divs.append(ol.ref_est_err(ref_len, int(g["qlen"][r]), regs[r], bs[r], mp) if len(regs[r]) and len(mp) else regs[r])
    alld = np.concatenate(divs) if divs else allr
    coff = np.concatenate([[0], np.cumsum([len(u) for u in us])]).astype(np.int64)
    boff = np.concatenate([[0], np.cumsum([len(b) for b in bs])]).astype(np.int64)
    out = os.path.join(OUT, os.path.basename(path))
    np.savez_compressed(out, min_cnt=np.int32(pv[7]), hash=hash_, qlen=g["qlen"].astype(np.int32), ref_len=ref_len, chains_off=coff,
                        u=np.concatenate(us) if coff[-1] else np.zeros(0, np.uint64), b_off=boff,
                        b=np.concatenate(bs) if boff[-1] else np.zeros((0, 2), np.uint64),
                        regs=allr.view(np.uint8).reshape(-1, 80), regs_div=alld.view(np.uint8).reshape(-1, 80))
    print(f"{os.path.basename(out)}: {R} reads, {coff[-1]} hits, {int((alld['div'] >= 0).sum())} with a divergence estimate")
