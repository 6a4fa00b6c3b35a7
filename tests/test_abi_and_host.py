"""CPU tier: the C-ABI library loads and exports every symbol the headers declare (no compute calls
without a GPU), the packet structs have the reference's sizes, the product fails loudly without a
GPU, and host-side helpers (generator, presets) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from minimap2_chaindp_amd import anchorgen as ag, chaindp, fpga, params as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    """Function names declared in a C header (lines of the form `type name(...);`)."""
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:chaindp|fpga)_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(chaindp.LIB_PATH)
    for header in ("chaindp.h", "chaindp_fpga.h"):
        names = _declared(header)
        assert len(names) >= 10
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"
    assert set(chaindp.ABI_SYMBOLS) == set(_declared("chaindp.h"))
    # every entry point is bound with explicit argument types (a default-int binding truncates 64-bit handles)
    L = chaindp.lib()
    assert [n for n in chaindp.ABI_SYMBOLS if getattr(L, n).argtypes is None] == []
    assert set(fpga.DRIVER_SYMBOLS) <= set(_declared("chaindp_fpga.h"))


def test_driver_abi_is_the_references_eleven_symbols():
    # nm -u of the reference's objects: exactly these are called
    assert sorted(fpga.DRIVER_SYMBOLS) == sorted([
        "fpga_init", "fpga_finalize", "fpga_get_writebuf", "fpga_get_writebuf_thread", "fpga_writebuf_submit",
        "fpga_get_retbuf", "fpga_release_retbuf", "fpga_set_params", "fpga_load_index", "fpga_exit_block",
        "fpga_set_block"])


def test_packet_struct_sizes_match_reference():
    # reference main.c:296-302 asserts these are 64 bytes; struct new_seed is 24 (minimap.h:51-55)
    assert C.sizeof(fpga.PktHdr) == 64 and C.sizeof(fpga.PktTask) == 64 and C.sizeof(fpga.PktResult) == 64
    assert chaindp.SEED_DTYPE.itemsize == 24
    assert C.sizeof(P.ChainParams) == 28


@pytest.mark.skipif(chaindp.device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(chaindp.ChainDPError, match="no HIP device"):
        chaindp.Device(0)
    assert fpga.lib().fpga_init(0) != 0          # the driver refuses to come up; nothing computes on the CPU


def test_packet_roundtrip_builders():
    """build_task_packet / parse_result_packet are inverse to the layouts of map.c:286-324 and map.c:918-931."""
    reads = [(7, np.arange(10, dtype=np.uint64).reshape(5, 2)), (9, np.zeros((0, 2), np.uint64)),
             (11, np.arange(6, dtype=np.uint64).reshape(3, 2))]
    pkt = fpga.build_task_packet(reads, gap_ref=5000, gap_qry=4000, tid=3)
    hdr = fpga.PktHdr.from_buffer_copy(pkt[:64])
    assert hdr.num == 3 and hdr.type == fpga.PKT_ANCHORS and hdr.size == len(pkt) and hdr.tid == 3
    assert len(pkt) == 64 + (64 + 128) + 64 + (64 + 64)          # payloads padded to 64 B
    t0 = fpga.PktTask.from_buffer_copy(pkt[64:128])
    assert (t0.gap_ref, t0.gap_qry, t0.seednum, t0.read_id, t0.n_segs) == (5000, 4000, 5, 7, 1)
    # a synthetic result packet
    seeds = np.zeros(2, chaindp.SEED_DTYPE)
    seeds["f"] = [15, 30]
    res = fpga.build_result_packet_for_test(hdr, [(7, seeds, 0), (9, None, 1)])
    parsed = fpga.parse_result_packet(res)
    assert parsed[0][0] == 7 and parsed[0][1] == 0 and list(parsed[0][2]["f"]) == [15, 30]
    assert parsed[1][0] == 9 and parsed[1][1] == 1 and parsed[1][2] is None


def test_generator_is_deterministic_and_sorted():
    off1, a1 = ag.generate("ava-ont", n_reads=20, seed=3, threads=1)
    off2, a2 = ag.generate("ava-ont", n_reads=20, seed=3, threads=7)
    assert np.array_equal(off1, off2) and np.array_equal(a1, a2)
    off3, a3 = ag.generate("ava-ont", n_reads=5, seed=3, first_read=10)        # shard of the same job
    assert np.array_equal(a3, a1[off1[10]:off1[15]])
    for r in range(20):
        x = a1[off1[r]:off1[r + 1], 0]
        assert np.all(x[:-1] <= x[1:])
    assert 3000 < (off1[-1] / 20) < 9000                                        # ~5-6k anchors per 10 kb read
    offs, _ = ag.generate("skew", n_reads=300, seed=1)
    n = np.diff(offs)
    assert n.min() < 400 and n.max() > 20000                                    # 1e2..1e5, log-uniform


def test_presets_follow_options_c():
    assert P.preset("map-ont").astuple() == (5000, 5000, 500, 25, 40, 0, 1)     # options.c:29-34,95-96
    assert P.preset("ava-ont").astuple() == (10000, 10000, 500, 25, 100, 0, 1)  # options.c:84-87
    assert P.preset("ava-pb").bw == 2000                                        # options.c:92


def test_anchor_dump_roundtrip_and_reference_dumps(tmp_path):
    """dumps written by oracle/mt_dump.c (the reference's own sketch/index/collect_seed_hits on its
    test/*.fa) load, regroup into batches and survive a write/read round trip."""
    from minimap2_chaindp_amd import dump
    d = os.path.join(ROOT, "tests", "golden", "dumps")
    mt = dump.read_dump(os.path.join(d, "mt_orang_vs_human.mapont.dump"))
    assert len(mt) == 1 and mt[0][0].astuple() == (5000, 5000, 500, 25, 40, 0, 1) and mt[0][1] == 3 and mt[0][2].shape == (346, 2)
    inv = dump.read_dump(os.path.join(d, "q_inv_vs_t_inv.mapont.dump"))
    assert len(inv) == 2
    b = dump.batches(mt + inv)
    assert len(b) == 1 and list(b[0][2]) == [0, 346, 346 + inv[0][2].shape[0], 346 + inv[0][2].shape[0] + inv[1][2].shape[0]]
    p = tmp_path / "x.dump"
    dump.write_dump(p, mt + inv)
    again = dump.read_dump(p)
    assert all(np.array_equal(x[2], y[2]) and x[0].astuple() == y[0].astuple() for x, y in zip(mt + inv, again))
