"""GPU tier: the reference PROGRAM, unmodified, with this repository's library linked in the place of libfpgadrv.a
(oracle/_ref/minimap2_chaindp, built in the build container by `make -C oracle ref-prog` from the reference's sources where
they lie; the executable travels with the snapshot).  It sketches, indexes, streams its index image through
fpga_load_index, sends its own minimizer packets, and turns the returned new_seed[] into alignments -- run on the
reference's own test FASTA (tests/golden/fa/, data files of the reference's test directory).

Expected: the chain SURVEY section 6 measured with the reference's chain.c on this pair (MT-orang -> MT-human, map-ont:
346 anchors, one chain of score 3189 over 342 anchors) shows up in the PAF line as s1:i:3189 cm:i:342."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = os.path.join(ROOT, "oracle", "_ref", "minimap2_chaindp")
FA = os.path.join(ROOT, "tests", "golden", "fa")


def _run(args, timeout=180):
    return subprocess.run([PROG] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)


@pytest.mark.skipif(not os.path.exists(PROG), reason="reference program not built (needs /root/reference: make -C oracle ref-prog)")
def test_reference_program_maps_through_the_gpu():
    # the reference hands n_threads - 10 threads to its mapping workers (map.c:711), so -t must exceed 10
    r = _run(["-x", "map-ont", "-t", "12", os.path.join(FA, "MT-human.fa"), os.path.join(FA, "MT-orang.fa")])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln.split("\t") for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) >= 1, r.stderr[-2000:]
    best = lines[0]
    assert best[0] == "MT_orang" and best[5] == "MT_human" and int(best[1]) == 16499 and int(best[6]) == 16569
    tags = dict(t.split(":", 2)[::2] for t in best[12:])
    assert tags.get("s1") == "3189" and tags.get("cm") == "342", best
    assert int(best[10]) > 15000                      # alignment block length: the whole mitochondrial genome


@pytest.mark.skipif(not os.path.exists(PROG), reason="reference program not built (needs /root/reference: make -C oracle ref-prog)")
def test_reference_program_inversion_pair():
    r = _run(["-x", "map-ont", "-t", "12", os.path.join(FA, "t-inv.fa"), os.path.join(FA, "q-inv.fa")])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln.split("\t") for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) >= 2 and {ln[4] for ln in lines} == {"+", "-"}, r.stdout      # both strands: the inversion is found
