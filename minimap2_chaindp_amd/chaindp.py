"""Python front end of the C ABI in include/chaindp.h (libchaindp_hip.so).

Thin by design: numpy arrays in, numpy arrays out, every call goes through the C ABI to the HIP
kernels.  There is NO CPU implementation behind this module: a missing library or a missing GPU
raises.  Per read the results equal the reference's mm_chain_dp_fpga (chain.c:218-327).
"""
import collections
import ctypes as C
import os

import numpy as np

from .params import ChainParams

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHAINDP_LIB: an alternative build of the same library (A/B timing of kernel variants on one box, tools/ab.sh)
LIB_PATH = os.environ.get("CHAINDP_LIB") or os.path.join(_HERE, "csrc", "libchaindp_hip.so")

SEED_DTYPE = np.dtype([("x", "<u8"), ("y", "<u8"), ("p", "<i4"), ("f", "<i4")])  # struct new_seed (minimap.h:51-55)

# every symbol include/chaindp.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "chaindp_device_count", "chaindp_create", "chaindp_destroy", "chaindp_last_error", "chaindp_chain_batch",
    "chaindp_upload", "chaindp_run", "chaindp_sync", "chaindp_download", "chaindp_compact",
    "chaindp_upload_gather", "chaindp_compact_offsets", "chaindp_download_seeds", "chaindp_host_alloc",
    "chaindp_host_free", "chaindp_run_device", "chaindp_set_profiling", "chaindp_get_kernel_ms",
    "chaindp_get_stats", "chaindp_set_ring", "chaindp_run_full", "chaindp_set_variant", "chaindp_upload_gather_ex", "chaindp_scatter_seeds", "chaindp_backtrack",
    "chaindp_index_create", "chaindp_index_destroy", "chaindp_collect_seeds", "chaindp_download_mini_pos", "chaindp_download_anchors", "chaindp_collect_seeds_gather", "chaindp_scatter_mini_pos",
    "chaindp_gen_regs", "chaindp_est_err",
    "chaindp_pipe_create", "chaindp_pipe_destroy", "chaindp_pipe_submit", "chaindp_pipe_wait", "chaindp_pipe_release",
    "chaindp_pipe_last_error", "chaindp_map_batch",
)

# chaindp_reg_t == mm_reg1_t (minimap.h:100-115), 80 bytes; `bits` is the bit-field word (rev = bit 10)
REG_DTYPE = np.dtype([(k, "<i4") for k in ("id", "cnt", "rid", "score", "qs", "qe", "rs", "re", "parent", "subsc", "as", "mlen", "blen", "n_sub", "score0")]
                     + [("bits", "<u4"), ("hash", "<u4"), ("div", "<f4"), ("reserved", "<u4", (2,))])


class ChainDPError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ChainDPError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                               "(make -C minimap2_chaindp_amd/csrc); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        P = C.POINTER(ChainParams)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
        L.chaindp_device_count.restype = i32
        L.chaindp_create.restype = vp
        L.chaindp_create.argtypes = [i32, i64, i64]
        L.chaindp_destroy.argtypes = [vp]
        L.chaindp_last_error.restype = C.c_char_p
        L.chaindp_last_error.argtypes = [vp]
        L.chaindp_chain_batch.argtypes = [vp, P, i64, vp, vp, vp, vp, vp, vp]
        L.chaindp_upload.argtypes = [vp, i64, vp, vp, vp]
        L.chaindp_run.argtypes = [vp, P]
        L.chaindp_run_full.argtypes = [vp, P]
        L.chaindp_sync.argtypes = [vp]
        L.chaindp_download.argtypes = [vp, vp, vp, vp]
        L.chaindp_compact.argtypes = [vp, P, vp, vp]
        L.chaindp_upload_gather.argtypes = [vp, i64, vp, vp, vp]
        L.chaindp_compact_offsets.argtypes = [vp, P, vp]
        L.chaindp_download_seeds.argtypes = [vp, i64, i64, vp]
        L.chaindp_host_alloc.restype = vp
        L.chaindp_host_alloc.argtypes = [C.c_size_t]
        L.chaindp_host_free.argtypes = [vp]
        L.chaindp_run_device.argtypes = [vp, P, i64, i64, vp, vp, vp, vp, vp, vp, vp]
        L.chaindp_set_profiling.argtypes = [vp, i32]
        L.chaindp_get_kernel_ms.argtypes = [vp, vp, vp, i32]
        L.chaindp_get_stats.argtypes = [vp, vp]
        L.chaindp_set_ring.argtypes = [vp, i32]
        L.chaindp_set_variant.argtypes = [vp, i32]
        L.chaindp_backtrack.argtypes = [vp, P, i32, vp, vp, vp, vp]
        L.chaindp_gen_regs.argtypes = [vp, vp, vp, vp]
        L.chaindp_est_err.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp]
        L.chaindp_index_create.restype = vp
        L.chaindp_index_create.argtypes = [i32, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t]
        L.chaindp_index_destroy.restype = None
        L.chaindp_index_destroy.argtypes = [vp]
        L.chaindp_collect_seeds.argtypes = [vp, vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp]
        L.chaindp_download_mini_pos.argtypes = [vp, vp]
        L.chaindp_collect_seeds_gather.argtypes = [vp, vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp]
        L.chaindp_scatter_mini_pos.argtypes = [vp, i64, vp]
        L.chaindp_upload_gather_ex.argtypes = [vp, i64, vp, vp, vp, i32]
        L.chaindp_scatter_seeds.argtypes = [vp, i64, vp]
        L.chaindp_device_count.argtypes = []
        L.chaindp_download_anchors.argtypes = [vp, vp]
        L.chaindp_map_batch.argtypes = [vp, vp, i32, i32, P, i32, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp]
        L.chaindp_pipe_create.restype = vp
        L.chaindp_pipe_create.argtypes = [i32, i32, i64, i64]
        L.chaindp_pipe_destroy.restype = None
        L.chaindp_pipe_destroy.argtypes = [vp]
        L.chaindp_pipe_submit.argtypes = [vp, P, i64, vp, vp, vp, i64]
        L.chaindp_pipe_wait.argtypes = [vp, vp]
        L.chaindp_pipe_release.argtypes = [vp]
        L.chaindp_pipe_last_error.restype = C.c_char_p
        L.chaindp_pipe_last_error.argtypes = [vp]
        _lib = L
    return _lib


def device_count():
    return lib().chaindp_device_count()


def _ptr(a):
    return None if a is None else a.ctypes.data


class Device:
    """One chaining context on one GPU (chaindp_ctx_t)."""

    def __init__(self, device=0, max_anchors=1 << 24, max_reads=1 << 20, ring=None):
        self._lib = lib()
        if self._lib.chaindp_device_count() <= 0:
            raise ChainDPError("no HIP device visible: the chaining DP runs only on the GPU (no CPU fallback)")
        self._ctx = self._lib.chaindp_create(device, max_anchors, max_reads)
        if not self._ctx:
            raise ChainDPError(self._lib.chaindp_last_error(None).decode())
        self.device = device
        self.max_anchors, self.max_reads = max_anchors, max_reads
        self._n_reads = self._total = 0
        self._indexes = []
        if ring is not None:
            self._check(self._lib.chaindp_set_ring(self._ctx, ring))

    # -- lifecycle
    def close(self):
        if self._ctx:
            for h in self._indexes:
                self._lib.chaindp_index_destroy(h)
            self._indexes = []
            self._lib.chaindp_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ChainDPError(f"chaindp error {rc}: {self._lib.chaindp_last_error(self._ctx).decode()}")

    @staticmethod
    def _prep(off, anchors, n_segs):
        off = np.ascontiguousarray(off, np.int64)
        anchors = np.ascontiguousarray(anchors, np.uint64).reshape(-1, 2)
        if len(off) < 1 or off[0] != 0 or int(off[-1]) != anchors.shape[0]:
            raise ValueError("off must start at 0 and end at the number of anchors")
        ns = None if n_segs is None else np.ascontiguousarray(n_segs, np.int32)
        if ns is not None and len(ns) != len(off) - 1:
            raise ValueError("n_segs must have one entry per read")
        return off, anchors, ns

    # -- host-buffer path
    def chain_batch(self, par, off, anchors, n_segs=None, want_v=True):
        """f, p, v (int32[total]) of every read of the batch; mm_chain_dp_fpga's arrays (chain.c:246-284)."""
        off, anchors, ns = self._prep(off, anchors, n_segs)
        tot = anchors.shape[0]
        f, p = np.empty(tot, np.int32), np.empty(tot, np.int32)
        v = np.empty(tot, np.int32) if want_v else None
        self._check(self._lib.chaindp_chain_batch(self._ctx, C.byref(par), len(off) - 1, _ptr(off), _ptr(anchors), _ptr(ns),
                                                  _ptr(f), _ptr(p), _ptr(v)))
        self._n_reads, self._total = len(off) - 1, tot
        return f, p, v

    def upload(self, off, anchors, n_segs=None):
        off, anchors, ns = self._prep(off, anchors, n_segs)
        self._check(self._lib.chaindp_upload(self._ctx, len(off) - 1, _ptr(off), _ptr(anchors), _ptr(ns)))
        self._n_reads, self._total = len(off) - 1, anchors.shape[0]

    def run(self, par):
        self._check(self._lib.chaindp_run(self._ctx, C.byref(par)))

    def run_full(self, par):
        """run() + the compaction kernels, asynchronous, results stay in HBM (the benchmark's step)."""
        self._check(self._lib.chaindp_run_full(self._ctx, C.byref(par)))

    def sync(self):
        self._check(self._lib.chaindp_sync(self._ctx))

    def download(self, want_v=True):
        f, p = np.empty(self._total, np.int32), np.empty(self._total, np.int32)
        v = np.empty(self._total, np.int32) if want_v else None
        self._check(self._lib.chaindp_download(self._ctx, _ptr(f), _ptr(p), _ptr(v)))
        return f, p, v

    def compact(self, par):
        """new_seed[] of every read (chain.c:286-317): (seeds_off int64[n_reads+1], seeds SEED_DTYPE[...])."""
        soff = np.zeros(self._n_reads + 1, np.int64)
        seeds = np.zeros(max(self._total, 1), SEED_DTYPE)
        self._check(self._lib.chaindp_compact(self._ctx, C.byref(par), _ptr(soff), _ptr(seeds)))
        return soff, seeds[:int(soff[-1])]

    def backtrack(self, par, min_cnt=3):
        """mm_chain_dp_bottom (chain.c:329-431) of every read on the GPU, after compact()/run_full():
        (chains_off, u uint64[...], b_off, b uint64[...,2])."""
        coff = np.zeros(self._n_reads + 1, np.int64)
        boff = np.zeros(self._n_reads + 1, np.int64)
        cap = max(self._total, 1)
        u = np.zeros(cap, np.uint64)
        b = np.zeros((cap, 2), np.uint64)
        self._check(self._lib.chaindp_backtrack(self._ctx, C.byref(par), min_cnt, _ptr(coff), _ptr(u), _ptr(boff), _ptr(b)))
        return coff, u[:int(coff[-1])], boff, b[:int(boff[-1])]

    # -- chains to hits (mm_gen_regs, hit.c:52-95; mm_est_err, esterr.c:30-64), after backtrack()
    def gen_regs(self, hash_, qlen, n_chains):
        """hash_ uint32[n_reads], qlen int32[n_reads] -> REG_DTYPE[n_chains] (n_chains = chains_off[-1] of backtrack())."""
        hash_ = np.ascontiguousarray(hash_, np.uint32)
        qlen = np.ascontiguousarray(qlen, np.int32)
        regs = np.zeros(max(int(n_chains), 1), REG_DTYPE)
        self._check(self._lib.chaindp_gen_regs(self._ctx, _ptr(hash_), _ptr(qlen), _ptr(regs)))
        return regs[:int(n_chains)]

    def est_err(self, regs_off, regs, qlen, ref_len, mini_pos_off=None, mini_pos=None):
        """mm_est_err on the hits `regs` (REG_DTYPE, read r owns regs_off[r]:regs_off[r+1]) -> (regs with div, n_match, n_tot).
        mini_pos_off / mini_pos None: the minimizer positions collect_seeds() left on the device."""
        regs_off = np.ascontiguousarray(regs_off, np.int64)
        regs = np.ascontiguousarray(regs, REG_DTYPE).copy()
        qlen = np.ascontiguousarray(qlen, np.int32)
        ref_len = np.ascontiguousarray(ref_len, np.int32)
        mt = np.zeros((max(len(regs), 1), 2), np.int32)
        mpo = None if mini_pos_off is None else np.ascontiguousarray(mini_pos_off, np.int64)
        mp = None if mini_pos is None else np.ascontiguousarray(mini_pos, np.uint64)
        self._check(self._lib.chaindp_est_err(self._ctx, _ptr(regs_off), _ptr(regs) if len(regs) else None, _ptr(qlen), _ptr(ref_len) if len(ref_len) else None,
                                              len(ref_len), None if mpo is None else _ptr(mpo), None if mp is None or not len(mp) else _ptr(mp), _ptr(mt)))
        return regs, mt[:len(regs), 0].copy(), mt[:len(regs), 1].copy()

    # -- seed collection on the GPU (collect_seed_hits, map.c:187-236, over the FPGA index image)
    def load_index(self, img):
        """img: the four blobs B, H, V, P of the reference's index image (index.c:603-720) -> a handle for collect_seeds."""
        blobs = [np.ascontiguousarray(b, np.uint8) for b in img]
        h = self._lib.chaindp_index_create(self.device, *sum(([_ptr(b) if b.size else None, int(b.size)] for b in blobs), []))
        if not h:
            raise ChainDPError((self._lib.chaindp_last_error(None) or b"").decode())
        self._indexes.append(h)
        return h

    def collect_seeds(self, index, flag, max_occ, mini_off, mini, bid, qlen, n_segs=None):
        """Minimizers of a batch -> sorted anchors resident on the device (as after upload()).  Returns (off int64[n_reads+1],
        anchors uint64[n,2], rep_len int32[n_reads], mini_pos_off, mini_pos uint64[...])."""
        mini_off = np.ascontiguousarray(mini_off, np.int64)
        n_reads = len(mini_off) - 1
        mini = np.ascontiguousarray(mini, np.uint64).reshape(-1, 2)
        bid = np.ascontiguousarray(bid, np.uint32); qlen = np.ascontiguousarray(qlen, np.int32)
        off = np.zeros(n_reads + 1, np.int64); mpo = np.zeros(n_reads + 1, np.int64); rep = np.zeros(max(n_reads, 1), np.int32)
        ns = None if n_segs is None else np.ascontiguousarray(n_segs, np.int32)
        self._check(self._lib.chaindp_collect_seeds(self._ctx, index, int(flag), int(max_occ), n_reads, _ptr(mini_off), _ptr(mini), _ptr(bid),
                                                    _ptr(qlen), _ptr(ns), _ptr(off), _ptr(rep), _ptr(mpo)))
        self._n_reads, self._total = n_reads, int(off[-1])
        a = np.zeros((max(self._total, 1), 2), np.uint64)
        self._check(self._lib.chaindp_download_anchors(self._ctx, _ptr(a)))
        mp = np.zeros(max(int(mpo[-1]), 1), np.uint64)
        self._check(self._lib.chaindp_download_mini_pos(self._ctx, _ptr(mp)))
        return off, a[:self._total], rep[:n_reads], mpo, mp[:int(mpo[-1])]

    def map_batch(self, index, flag, max_occ, par, min_cnt, mini_off, mini, bid, qlen, hash_, regs_cap=None):
        """Minimizers in, hits out, everything in between resident (chaindp_map_batch): (regs_off int64[n_reads+1], regs REG_DTYPE[...],
        rep_len int32[n_reads], n_anchors)."""
        mini_off = np.ascontiguousarray(mini_off, np.int64)
        n_reads = len(mini_off) - 1
        mini = np.ascontiguousarray(mini, np.uint64).reshape(-1, 2)
        bid = np.ascontiguousarray(bid, np.uint32); qlen = np.ascontiguousarray(qlen, np.int32); hash_ = np.ascontiguousarray(hash_, np.uint32)
        cap = int(regs_cap) if regs_cap is not None else max(len(mini) // 4, 1024)
        roff = np.zeros(n_reads + 1, np.int64); rep = np.zeros(max(n_reads, 1), np.int32)
        regs = np.zeros(max(cap, 1), REG_DTYPE)
        na = C.c_int64(0)
        rc = self._lib.chaindp_map_batch(self._ctx, index, int(flag), int(max_occ), C.byref(par), int(min_cnt), n_reads, _ptr(mini_off), _ptr(mini),
                                         _ptr(bid), _ptr(qlen), _ptr(hash_), _ptr(roff), _ptr(regs), cap, _ptr(rep), C.byref(na))
        if rc == -2 and int(roff[-1]) > cap:                                   # more hits than guessed: they are resident, fetch them
            regs = self.gen_regs(hash_, qlen, int(roff[-1]))
        else:
            self._check(rc)
        self._n_reads, self._total = n_reads, int(na.value)
        return roff, regs[:int(roff[-1])], rep[:n_reads], int(na.value)

    # -- device-pointer path (torch tensors or any other HBM allocation)
    def run_device(self, par, n_reads, total, d_off, d_a, d_n_segs, d_f, d_p, d_v, stream=0):
        self._check(self._lib.chaindp_run_device(self._ctx, C.byref(par), n_reads, total, d_off, d_a, d_n_segs or None,
                                                 d_f, d_p, d_v, stream or None))

    def set_ring(self, ring):
        self._check(self._lib.chaindp_set_ring(self._ctx, ring))

    def set_variant(self, force_general):
        """0 / False: two units per wave (k_chain_twin) + one per wave for what it hands over; 1 / True: everything through the
        general variant of k_chain_units; 2: k_chain_units only (table-driven where it applies)."""
        self._check(self._lib.chaindp_set_variant(self._ctx, int(force_general)))

    # -- measurement
    def set_profiling(self, on=True):
        self._check(self._lib.chaindp_set_profiling(self._ctx, int(bool(on))))

    def kernel_ms(self, reset=False):
        """Accumulated HIP-event device time per kernel: dict name -> (ms, launches)."""
        ms = (C.c_double * 4)()
        n = (C.c_int64 * 4)()
        self._check(self._lib.chaindp_get_kernel_ms(self._ctx, ms, n, int(reset)))
        return {k: (ms[i], n[i]) for i, k in enumerate(("prepass", "chain_dp", "compact", "backtrack"))}

    def leftover_units(self):
        """Units the two-per-wave kernel handed over to the one-per-wave kernel in the last run (test / tuning hook)."""
        self._lib.chaindp_debug_leftover.restype = C.c_int64
        self._lib.chaindp_debug_leftover.argtypes = [C.c_void_p]
        return int(self._lib.chaindp_debug_leftover(self._ctx))

    def deep_units(self):
        """Units the one-per-wave kernel handed over to k_chain_dense in the last run (test / tuning hook)."""
        self._lib.chaindp_debug_deep_units.restype = C.c_int64
        self._lib.chaindp_debug_deep_units.argtypes = [C.c_void_p]
        return int(self._lib.chaindp_debug_deep_units(self._ctx))

    def set_quad(self, on=True):
        """Test / A-B hook: let k_chain_quad (four units per wave; off by default: measured slower) take the batches it can."""
        self._lib.chaindp_debug_set_quad.restype = C.c_int
        self._lib.chaindp_debug_set_quad.argtypes = [C.c_void_p, C.c_int]
        self._check(self._lib.chaindp_debug_set_quad(self._ctx, int(bool(on))))

    def quad_took(self):
        """True if k_chain_quad took the last batch (test hook)."""
        self._lib.chaindp_debug_quad_took.restype = C.c_int
        self._lib.chaindp_debug_quad_took.argtypes = [C.c_void_p]
        return int(self._lib.chaindp_debug_quad_took(self._ctx)) == 1

    def set_twin_handover(self, mode=0):
        """Test hook: what k_chain_twin hands over to k_chain_units whatever the units look like -- 0 nothing extra, 1 every unit
        untouched, 2 every unit after its first 64-anchor tile (k_chain_units resumes behind the flushed tiles)."""
        self._lib.chaindp_debug_set_twin_handover.restype = C.c_int
        self._lib.chaindp_debug_set_twin_handover.argtypes = [C.c_void_p, C.c_int]
        self._check(self._lib.chaindp_debug_set_twin_handover(self._ctx, int(mode)))

    def set_deep_handover(self, on=True):
        """Test hook: False keeps every unit in the launch that took it (k_chain_units then serves long scans from HBM/L2); 2 hands
        over any unit with a few deep scans, whatever its length, to k_chain_dense, 3 to k_chain_dense1, 4 to k_chain_dense16 (small
        test inputs reach every kernel)."""
        self._lib.chaindp_debug_set_deep_handover.restype = C.c_int
        self._lib.chaindp_debug_set_deep_handover.argtypes = [C.c_void_p, C.c_int]
        self._check(self._lib.chaindp_debug_set_deep_handover(self._ctx, on if on in (2, 3, 4) else int(bool(on))))

    def stats(self):
        st = (C.c_int64 * 4)()
        self._check(self._lib.chaindp_get_stats(self._ctx, st))
        return dict(units=st[0], singletons=st[1], anchors=st[2], reads=st[3])


class PinnedArray:
    """A numpy view of pinned (DMA-able) host memory from chaindp_host_alloc; what makes Pipe's copies asynchronous."""

    def __init__(self, shape, dtype):
        self._lib = lib()
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        self._ptr = self._lib.chaindp_host_alloc(max(n, 1))
        if not self._ptr:
            raise ChainDPError("chaindp_host_alloc failed (no GPU runtime or out of pinned memory)")
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(n, 1)).from_address(self._ptr))[:n].view(dtype).reshape(shape)

    def free(self):
        if self._ptr:
            self.array = None
            self._lib.chaindp_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PipeResult(C.Structure):
    _fields_ = [("tag", C.c_int64), ("n_reads", C.c_int64), ("n_anchors", C.c_int64), ("n_seeds", C.c_int64),
                ("seeds_off", C.c_void_p), ("seeds", C.c_void_p)]


class Pipe:
    """chaindp_pipe_t: batches stream through `depth` contexts so that H2D(n+1), kernels(n) and D2H(n-1) overlap."""

    def __init__(self, device=0, depth=3, max_anchors=1 << 24, max_reads=1 << 20):
        self._lib = lib()
        if self._lib.chaindp_device_count() <= 0:
            raise ChainDPError("no HIP device visible: the chaining DP runs only on the GPU (no CPU fallback)")
        self._p = self._lib.chaindp_pipe_create(device, depth, max_anchors, max_reads)
        if not self._p:
            raise ChainDPError((self._lib.chaindp_pipe_last_error(None) or b"").decode())
        self.depth = depth
        self._keep = collections.deque()       # host arrays of the batches in flight, oldest first (batches complete in submission order)

    def close(self):
        if self._p:
            self._lib.chaindp_pipe_destroy(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise ChainDPError(f"chaindp pipe error {rc}: {self._lib.chaindp_pipe_last_error(self._p).decode()}")

    def submit(self, par, off, anchors, n_segs=None, tag=0):
        """Asynchronous.  off / anchors / n_segs must stay alive (they are kept referenced here) until wait() returns the batch."""
        off, anchors, ns = Device._prep(off, anchors, n_segs)
        rc = self._lib.chaindp_pipe_submit(self._p, C.byref(par), len(off) - 1, _ptr(off), _ptr(anchors), _ptr(ns), tag)
        if rc == -5:
            return False
        self._check(rc)
        self._keep.append((off, anchors, ns))   # by submission, not by tag: two batches may carry the same tag
        return True

    def wait(self, copy=True):
        """Oldest batch: (tag, seeds_off, seeds).  With copy=False the arrays alias the pipe's pinned buffers and are valid
        until release()."""
        r = _PipeResult()
        self._check(self._lib.chaindp_pipe_wait(self._p, C.byref(r)))
        soff = np.ctypeslib.as_array((C.c_int64 * (r.n_reads + 1)).from_address(r.seeds_off))
        seeds = np.ctypeslib.as_array((C.c_uint8 * max(r.n_seeds * 24, 1)).from_address(r.seeds))[:r.n_seeds * 24].view(SEED_DTYPE)
        if copy:
            soff, seeds = soff.copy(), seeds.copy()
        if self._keep:
            self._keep.popleft()
        return r.tag, soff, seeds

    def release(self):
        self._check(self._lib.chaindp_pipe_release(self._p))
