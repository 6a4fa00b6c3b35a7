"""Host-side mirror of the reference's accelerator driver ABI (fpga.h:37-62) as served by
libchaindp_hip.so (include/chaindp_fpga.h): ctypes declarations, the packet layouts of
fpga_chaindp.h:46-87, builders/parsers that follow map.c:286-324 (package_task) and map.c:918-931
(result walk), and a small client that plays the reference's producer threads and its single
receiver (recv_task_thread, fpga_chaindp.c:228) for tests and demos.
"""
import ctypes as C
import threading

import numpy as np

from . import chaindp
from .chaindp import SEED_DTYPE

PKT_MINIMIZERS = 3        # the reference's task packets (map.c:302)
PKT_ANCHORS = 0x41        # this build's: payload = the read's sorted anchors

SHIM_EXTRA_SYMBOLS = ("chaindp_fpga_configure", "chaindp_fpga_configure_capacity", "chaindp_fpga_configure_groups", "chaindp_fpga_configure_services", "chaindp_fpga_stats",
                      "chaindp_fpga_stats_gpu")

DRIVER_SYMBOLS = (
    "fpga_init", "fpga_finalize", "fpga_get_retbuf", "fpga_release_retbuf", "fpga_get_writebuf",
    "fpga_get_writebuf_thread", "fpga_writebuf_submit", "fpga_exit_block", "fpga_set_block",
    "fpga_set_params", "fpga_load_index",
)


class PktHdr(C.Structure):              # chaindp_sndhdr_t, fpga_chaindp.h:79-87
    _pack_ = 1
    _fields_ = [("magic", C.c_uint32), ("size", C.c_uint32), ("tid", C.c_uint16), ("num", C.c_uint16),
                ("type", C.c_uint8), ("lat", C.c_uint8), ("reserve1", C.c_uint8 * 50)]


class PktTask(C.Structure):             # collect_task_t, fpga_chaindp.h:46-58
    _pack_ = 1
    _fields_ = [("gap_qry", C.c_int32), ("gap_ref", C.c_int32), ("seednum", C.c_int32), ("qlensum", C.c_int32),
                ("read_id", C.c_uint32), ("bid", C.c_uint32), ("n_segs", C.c_int16), ("b", C.c_char),
                ("reserve1", C.c_char * 1), ("mv_a", C.c_uint64), ("reserve2", C.c_char * 28)]


class PktResult(C.Structure):           # collect_result_t, fpga_chaindp.h:60-68
    _pack_ = 1
    _fields_ = [("err_flag", C.c_uint32), ("read_id", C.c_uint32), ("sub_size", C.c_uint32), ("n_a", C.c_uint32),
                ("n_minipos", C.c_uint32), ("rep_len", C.c_uint32), ("reserve1", C.c_char * 40)]


def _align64(n):
    return (n + 63) & ~63


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = chaindp.lib()
        vp = C.c_void_p
        L.fpga_init.restype = C.c_int
        L.fpga_init.argtypes = [C.c_int]
        L.fpga_finalize.restype = None
        L.fpga_get_retbuf.restype = vp
        L.fpga_get_retbuf.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.fpga_release_retbuf.argtypes = [vp]
        L.fpga_get_writebuf.restype = vp
        L.fpga_get_writebuf.argtypes = [C.c_ulong, C.c_int]
        L.fpga_get_writebuf_thread.restype = vp
        L.fpga_get_writebuf_thread.argtypes = [C.c_ulong, C.c_int, C.c_int]
        L.fpga_writebuf_submit.argtypes = [vp, C.c_uint, C.c_uint]
        L.fpga_exit_block.restype = None
        L.fpga_set_block.restype = None
        L.fpga_set_params.restype = None
        L.fpga_set_params.argtypes = [C.c_int] * 6
        L.fpga_load_index.restype = None
        L.fpga_load_index.argtypes = [vp, C.c_int, C.c_int]
        L.chaindp_fpga_configure.restype = None
        L.chaindp_fpga_configure.argtypes = [C.c_int, C.c_int, C.c_ulong]
        L.chaindp_fpga_stats.restype = None
        L.chaindp_fpga_stats.argtypes = [vp]
        L.chaindp_fpga_configure_capacity.restype = None
        L.chaindp_fpga_configure_capacity.argtypes = [C.c_int64, C.c_int64]
        L.chaindp_fpga_configure_services.restype = None
        L.chaindp_fpga_configure_services.argtypes = [C.c_int]
        L.chaindp_fpga_configure_groups.restype = None
        L.chaindp_fpga_configure_groups.argtypes = [C.c_int]
        L.chaindp_fpga_stats_gpu.restype = C.c_int
        L.chaindp_fpga_stats_gpu.argtypes = [C.c_int, vp]
        _lib = L
    return _lib


def build_task_packet(reads, gap_ref, gap_qry, tid=0, n_segs=1, pkt_type=PKT_ANCHORS, qlensum=0):
    """reads: list of (read_id, payload uint64[n,2]) or (read_id, payload, bid, qlen): anchors for PKT_ANCHORS, the read's
    minimizers for PKT_MINIMIZERS (then bid and qlen matter, map.c:350,523).  Layout of package_task (map.c:286-324)."""
    body = bytearray()
    for item in reads:
        read_id, a = item[0], item[1]
        bid, qlen = (item[2], item[3]) if len(item) > 2 else (0, qlensum)
        a = np.ascontiguousarray(a, np.uint64).reshape(-1, 2)
        t = PktTask()
        t.gap_qry, t.gap_ref, t.seednum, t.qlensum = gap_qry, gap_ref, a.shape[0], qlen
        t.read_id, t.bid, t.n_segs = read_id, bid, n_segs
        body += bytes(t)
        raw = a.tobytes()
        body += raw + b"\0" * (_align64(len(raw)) - len(raw))
    h = PktHdr()
    h.size, h.tid, h.num, h.type = 64 + len(body), tid, len(reads), pkt_type
    return bytes(h) + bytes(body)


def parse_result_packet(buf):
    """-> list of (read_id, err_flag, new_seed[] as SEED_DTYPE or None).  Walk of map.c:918-931,970-971."""
    buf = bytes(buf)
    h = PktHdr.from_buffer_copy(buf[:64])
    pos, out = 64, []
    for _ in range(h.num):
        r = PktResult.from_buffer_copy(buf[pos:pos + 64])
        pos += 64
        if r.err_flag == 1:
            out.append((r.read_id, 1, None))
            continue
        nbytes = r.n_a * SEED_DTYPE.itemsize
        seeds = np.frombuffer(buf, SEED_DTYPE, r.n_a, pos).copy()
        pos += _align64(nbytes) + _align64(r.n_minipos * 8)
        assert r.sub_size == 64 + _align64(nbytes) + _align64(r.n_minipos * 8)
        out.append((r.read_id, 0, seeds))
    assert pos == len(buf) or pos == h.size
    return out


def parse_result_packet_full(buf):
    """-> list of (read_id, err_flag, new_seed[], mini_pos uint64[], rep_len): parse_result_packet plus the by-products of
    seed collection that minimizer packets return (map.c:530-531,547-552)."""
    buf = bytes(buf)
    h = PktHdr.from_buffer_copy(buf[:64])
    pos, out = 64, []
    for _ in range(h.num):
        r = PktResult.from_buffer_copy(buf[pos:pos + 64])
        pos += 64
        if r.err_flag == 1:
            out.append((r.read_id, 1, None, None, 0))
            continue
        nbytes = r.n_a * SEED_DTYPE.itemsize
        seeds = np.frombuffer(buf, SEED_DTYPE, r.n_a, pos).copy()
        pos += _align64(nbytes)
        mini_pos = np.frombuffer(buf, np.uint64, r.n_minipos, pos).copy()
        pos += _align64(r.n_minipos * 8)
        out.append((r.read_id, 0, seeds, mini_pos, r.rep_len))
    return out


def load_index(img):
    """img: the four blobs B, H, V, P of the reference's index image (index.c:603-720), sent as main.c:201-204 does."""
    L = lib()
    for k, blob in enumerate(img):
        blob = np.ascontiguousarray(blob, np.uint8)
        if blob.size:
            L.fpga_load_index(blob.ctypes.data, int(blob.size), 4 + k)      # TYPE_INDEX_B.. (fpga.h:20-23)


def build_result_packet_for_test(hdr, items):
    """Inverse of parse_result_packet, only used to test the parser without a GPU."""
    body = bytearray()
    for read_id, seeds, err in items:
        r = PktResult()
        r.read_id, r.err_flag = read_id, err
        if err:
            r.sub_size = 64
            body += bytes(r)
            continue
        raw = np.ascontiguousarray(seeds).tobytes()
        r.n_a = len(seeds)
        r.sub_size = 64 + _align64(len(raw))
        body += bytes(r) + raw + b"\0" * (_align64(len(raw)) - len(raw))
    h = PktHdr.from_buffer_copy(bytes(hdr))
    h.size, h.num = 64 + len(body), len(items)
    return bytes(h) + bytes(body)


class Driver:
    """fpga_init ... fpga_finalize bracket (main.c:511-519,605-615) with a receiver thread that plays
    recv_task_thread (fpga_chaindp.c:228-270): blocks in fpga_get_retbuf, copies the packet, releases it."""

    def __init__(self, bw=500, is_cdna=0, max_skip=25, min_sc=40, n_gpus=0, max_packets_per_batch=64, flag=0, max_occ=0, index=None,
                 max_anchors_per_batch=32 << 20, max_reads_per_batch=1 << 19, n_groups=0):
        self.L = lib()
        self.L.chaindp_fpga_configure(n_gpus, max_packets_per_batch, 0)
        self.L.chaindp_fpga_configure_groups(n_groups)                     # 0: one service group per GPU
        self.L.chaindp_fpga_configure_capacity(max_anchors_per_batch, max_reads_per_batch)
        if self.L.fpga_init(0) != 0:
            raise chaindp.ChainDPError("fpga_init failed: no GPU (there is no CPU fallback)")
        if index is not None:
            load_index(index)                                                # main.c:201-204
        self.L.fpga_set_params(bw, is_cdna, max_skip, min_sc, flag, max_occ)   # main.c:243
        self.results = []
        self._lock = threading.Lock()
        self._rx = threading.Thread(target=self._recv_loop, daemon=True)
        self._rx.start()

    def _recv_loop(self):
        n = C.c_int(0)
        while True:
            p = self.L.fpga_get_retbuf(C.byref(n), 3)          # RET_TYPE_CS
            if n.value == 0:
                return                                          # fpga_exit_block (fpga_chaindp.c:242)
            data = C.string_at(p, n.value)
            self.L.fpga_release_retbuf(p)
            with self._lock:
                self.results.append(data)

    def submit(self, packet, tid=0):
        """map.c:439-444: get a driver buffer (retry while busy), copy the packet in, submit."""
        import time
        while True:
            buf = self.L.fpga_get_writebuf_thread(len(packet), 0, tid)   # BUF_TYPE_SW
            if buf:
                break
            time.sleep(50e-6)
        C.memmove(buf, packet, len(packet))
        return self.L.fpga_writebuf_submit(buf, len(packet), 1)          # TYPE_CD

    def wait_results(self, n_packets, timeout=120.0):
        import time
        t0 = time.time()
        while True:
            with self._lock:
                if len(self.results) >= n_packets:
                    return list(self.results)
            if time.time() - t0 > timeout:
                raise TimeoutError(f"{len(self.results)} of {n_packets} result packets after {timeout}s")
            time.sleep(1e-3)

    def stats(self):
        st = (C.c_int64 * 5)()
        self.L.chaindp_fpga_stats(st)
        return dict(packets=st[0], reads=st[1], anchors=st[2], batches=st[3], err_reads=st[4])

    def stats_gpu(self):
        """Per GPU in use: (device batches, anchors chained)."""
        st = (C.c_int64 * 2)()
        n = self.L.chaindp_fpga_stats_gpu(0, st)
        out = []
        for d in range(max(n, 0)):
            self.L.chaindp_fpga_stats_gpu(d, st)
            out.append((st[0], st[1]))
        return out

    def close(self):
        self.L.fpga_exit_block()                 # main.c:608
        self._rx.join(timeout=10)
        self.L.fpga_set_block()                  # main.c:613
        self.L.fpga_finalize()                   # main.c:614

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
