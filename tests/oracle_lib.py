"""ctypes access to oracle/liboracle.so (CPU restatement + wave model) and, when
built, oracle/_ref/*.so (the unmodified reference).  CHECKER ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmm2chain_ref.so")
CAP_SO = os.path.join(ROOT, "oracle", "_ref", "libmm2chain_cap.so")
SEED_SO = os.path.join(ROOT, "oracle", "libseedoracle.so")

SEED_DTYPE = np.dtype([("x", "<u8"), ("y", "<u8"), ("p", "<i4"), ("f", "<i4")])  # struct new_seed, 24 B
assert SEED_DTYPE.itemsize == 24
# mm_reg1_t (minimap.h:100-115), 80 B, with two reserved words where its mm_extra_t pointer is; `bits` is the bit-field word (rev = bit 10)
REG_DTYPE = np.dtype([(k, "<i4") for k in ("id", "cnt", "rid", "score", "qs", "qe", "rs", "re", "parent", "subsc", "as", "mlen", "blen", "n_sub", "score0")]
                     + [("bits", "<u4"), ("hash", "<u4"), ("div", "<f4"), ("reserved", "<u4", (2,))])
assert REG_DTYPE.itemsize == 80
REF_REG_BYTES = 80            # sizeof(mm_reg1_t) in the reference: 72 B of fields, then the mm_extra_t pointer


class CoParams(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("max_dist_x", "max_dist_y", "bw", "max_skip", "min_sc", "is_cdna", "n_segs")]


def _co(par):
    if isinstance(par, CoParams):
        return par
    return CoParams(*[getattr(par, k) for k, _ in CoParams._fields_])


_oracle = None
_cap = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(ORACLE_SO)
        P = C.POINTER(CoParams)
        lib.co_chain_fpv.restype = C.c_int64
        lib.co_chain_fpv.argtypes = [P, C.c_int64] + [C.c_void_p] * 5
        lib.co_compact.restype = C.c_uint32
        lib.co_compact.argtypes = [P, C.c_int64] + [C.c_void_p] * 6
        lib.co_chain_bottom.restype = C.c_void_p
        lib.co_chain_bottom.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        lib.co_batch_fpv.restype = C.c_int64
        lib.co_batch_fpv.argtypes = [P, C.c_int64] + [C.c_void_p] * 6 + [C.c_int]
        lib.co_time_top.restype = C.c_double
        lib.co_time_top.argtypes = [P, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
        lib.wm_batch_fpv.restype = C.c_int64
        lib.wm_batch_fpv.argtypes = [P, C.c_int64] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p]
        lib.co_radix_sort_128x.argtypes = [C.c_void_p, C.c_void_p]
        lib.co_radix_sort_64.argtypes = [C.c_void_p, C.c_void_p]
        lib.co_gen_regs.restype = None
        lib.co_gen_regs.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.co_est_err.restype = None
        lib.co_est_err.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO) and os.path.exists(CAP_SO)


def ref_cap():
    global _cap
    if _cap is None:
        lib = C.CDLL(CAP_SO)
        lib.ref_capture_top.restype = C.c_uint32
        lib.ref_capture_top.argtypes = [C.c_int] * 7 + [C.c_int64] + [C.c_void_p] * 5
        lib.mm_chain_dp_bottom.restype = C.c_void_p
        lib.mm_chain_dp_bottom.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                                           C.c_void_p, C.c_void_p, C.c_uint32]
        lib.radix_sort_128x.argtypes = [C.c_void_p, C.c_void_p]
        lib.radix_sort_64.argtypes = [C.c_void_p, C.c_void_p]
        _cap = lib
    return _cap


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        _ref.mm_gen_regs.restype = C.c_void_p
        _ref.mm_gen_regs.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _ref.mm_est_err.restype = None
        _ref.mm_est_err.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    return _ref


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


# ---- single-read calls -------------------------------------------------------

def oracle_fpv(par, a):
    """a: uint64[n,2].  Returns f,p,v int32[n] and the pair-evaluation count."""
    n = a.shape[0]
    f, p, v, t = (np.empty(max(n, 1), np.int32) for _ in range(4))
    ev = oracle().co_chain_fpv(C.byref(_co(par)), n, a.ctypes.data, f.ctypes.data, p.ctypes.data, v.ctypes.data, t.ctypes.data)
    return f[:n], p[:n], v[:n], ev


def oracle_compact(par, a, f, p, v):
    n = a.shape[0]
    out = np.zeros(max(n, 1), SEED_DTYPE)
    ids = np.empty(max(n, 1), np.int32)
    m = oracle().co_compact(C.byref(_co(par)), n, a.ctypes.data, f.ctypes.data, p.ctypes.data, v.ctypes.data,
                            out.ctypes.data, ids.ctypes.data)
    return out[:m].copy()


def oracle_bottom(min_cnt, min_sc, seeds):
    n_u = C.c_int(0)
    u = C.c_void_p(0)
    seeds = np.ascontiguousarray(seeds)
    b = oracle().co_chain_bottom(min_cnt, min_sc, seeds.ctypes.data, seeds.shape[0], C.byref(n_u), C.byref(u))
    return _take_bottom(b, u, n_u.value)


def _take_bottom(b, u, n_u):
    if not b:
        return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
    uu = np.ctypeslib.as_array(C.cast(u, C.POINTER(C.c_uint64)), (n_u,)).copy() if n_u else np.zeros(0, np.uint64)
    tot = int((uu & np.uint64(0xffffffff)).sum())
    bb = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint64)), (tot * 2,)).copy().reshape(tot, 2) if tot else np.zeros((0, 2), np.uint64)
    _libc.free(b)
    _libc.free(u)
    return uu, bb


def ref_fpv_seeds(par, a):
    """The unmodified reference mm_chain_dp_fpga on a[] -> (f, p, v, new_seed[])."""
    n = a.shape[0]
    f, p, v = (np.zeros(max(n, 1), np.int32) for _ in range(3))
    seeds = np.zeros(max(n, 1), SEED_DTYPE)
    a = np.ascontiguousarray(a)
    t = _co(par)
    m = ref_cap().ref_capture_top(t.max_dist_x, t.max_dist_y, t.bw, t.max_skip, t.min_sc, t.is_cdna, t.n_segs,
                                  n, a.ctypes.data, f.ctypes.data, p.ctypes.data, v.ctypes.data, seeds.ctypes.data)
    return f[:n], p[:n], v[:n], seeds[:m].copy()


def ref_bottom(min_cnt, min_sc, n_segs, seeds):
    n_u = C.c_int(0)
    u = C.c_void_p(0)
    seeds = np.ascontiguousarray(seeds)
    b = ref_cap().mm_chain_dp_bottom(min_cnt, min_sc, n_segs, C.byref(n_u), C.byref(u), None, seeds.ctypes.data, seeds.shape[0])
    return _take_bottom(b, u, n_u.value)


# ---- chains -> hits (hit.c:52-95) and their divergence estimate (esterr.c:30-64) ----

def oracle_gen_regs(hash_, qlen, u, b):
    u = np.ascontiguousarray(u, np.uint64)
    b = np.ascontiguousarray(b, np.uint64)
    out = np.zeros(len(u), REG_DTYPE)
    if len(u):
        oracle().co_gen_regs(hash_, qlen, len(u), u.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def oracle_est_err(ref_len, qlen, regs, b, mini_pos):
    regs = regs.copy()
    ref_len = np.ascontiguousarray(ref_len, np.int32)
    b = np.ascontiguousarray(b, np.uint64)
    mini_pos = np.ascontiguousarray(mini_pos, np.uint64)
    n_match, n_tot = np.zeros(max(len(regs), 1), np.int32), np.zeros(max(len(regs), 1), np.int32)
    oracle().co_est_err(ref_len.ctypes.data, qlen, len(regs), regs.ctypes.data, b.ctypes.data, len(mini_pos), mini_pos.ctypes.data,
                        n_match.ctypes.data, n_tot.ctypes.data)
    return regs, n_match[:len(regs)], n_tot[:len(regs)]


def _ref_regs_to_np(ptr, n):
    raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (n * REF_REG_BYTES,)).reshape(n, REF_REG_BYTES)
    out = np.zeros(n, REG_DTYPE)
    out.view(np.uint8).reshape(n, 80)[:, :72] = raw[:, :72]
    return out


def ref_gen_regs(hash_, qlen, u, b):
    """The unmodified reference mm_gen_regs (returns its calloc'd mm_reg1_t[] as REG_DTYPE records)."""
    u = np.ascontiguousarray(u, np.uint64)
    b = np.ascontiguousarray(b, np.uint64)
    if not len(u):
        return np.zeros(0, REG_DTYPE)
    ptr = ref().mm_gen_regs(None, hash_, qlen, len(u), u.ctypes.data, b.ctypes.data)
    out = _ref_regs_to_np(ptr, len(u))
    _libc.free(ptr)
    return out


class _RefIdxSeq(C.Structure):           # mm_idx_seq_t, minimap.h:58-62
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint64), ("len", C.c_uint32)]


class _RefIdx(C.Structure):              # the head of mm_idx_t, minimap.h:77-81 (mm_est_err reads seq[rid].len only)
    _fields_ = [("b", C.c_int32), ("w", C.c_int32), ("k", C.c_int32), ("flag", C.c_int32), ("n_seq", C.c_uint32), ("seq", C.POINTER(_RefIdxSeq))]


def ref_est_err(ref_len, qlen, regs, b, mini_pos):
    """The unmodified reference mm_est_err on mm_reg1_t records rebuilt from `regs`; returns the records with its div."""
    n = len(regs)
    raw = np.zeros((n, REF_REG_BYTES), np.uint8)
    raw[:, :72] = regs.view(np.uint8).reshape(n, 80)[:, :72]
    seqs = (_RefIdxSeq * max(len(ref_len), 1))()
    for i, L in enumerate(ref_len):
        seqs[i].len = int(L)
    idx = _RefIdx(14, 10, 15, 0, len(ref_len), C.cast(seqs, C.POINTER(_RefIdxSeq)))
    b = np.ascontiguousarray(b, np.uint64)
    mini_pos = np.ascontiguousarray(mini_pos, np.uint64)
    ref().mm_est_err(C.byref(idx), qlen, n, raw.ctypes.data, b.ctypes.data, len(mini_pos), mini_pos.ctypes.data)
    out = np.zeros(n, REG_DTYPE)
    out.view(np.uint8).reshape(n, 80)[:, :72] = raw[:, :72]
    return out


# ---- seed collection (collect_seed_hits, map.c:112-236, over the FPGA index image): oracle/seed_oracle.cpp ----

_seed = None


def seed_oracle():
    global _seed
    if _seed is None:
        lib = C.CDLL(SEED_SO)
        lib.so_index_create.restype = C.c_void_p
        lib.so_index_create.argtypes = [C.c_void_p, C.c_size_t] * 4
        lib.so_index_destroy.restype = None
        lib.so_index_destroy.argtypes = [C.c_void_p]
        lib.so_collect_seed_hits.restype = C.c_int
        lib.so_collect_seed_hits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                             C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int)]
        _seed = lib
    return _seed


class SeedIndex:
    """The reference's index image (blobs B, H, V, P of index.c:603-720) opened by the CPU restatement."""

    def __init__(self, blobs):
        self._blobs = [np.ascontiguousarray(b, np.uint8) for b in blobs]
        args = sum(([b.ctypes.data if b.size else None, int(b.size)] for b in self._blobs), [])
        self._h = seed_oracle().so_index_create(*args)
        if not self._h:
            raise ValueError("incomplete index image (B, H and V blobs are required)")

    def close(self):
        if self._h:
            seed_oracle().so_index_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def collect_seeds(self, flag, max_occ, bid, qlen, mini):
        """collect_seed_hits for one read -> (anchors uint64[n,2] sorted as radix_sort_128x leaves them, rep_len, mini_pos uint64[])."""
        mini = np.ascontiguousarray(mini, np.uint64).reshape(-1, 2)
        n, rl, nmp = C.c_int64(0), C.c_int(0), C.c_int(0)
        mp = np.zeros(len(mini) + 1, np.uint64)
        cap = 1 << 12
        while True:
            out = np.zeros((cap, 2), np.uint64)
            rc = seed_oracle().so_collect_seed_hits(self._h, int(flag), int(max_occ), int(bid), int(qlen), mini.ctypes.data, len(mini),
                                                    out.ctypes.data, cap, C.byref(n), C.byref(rl), mp.ctypes.data, C.byref(nmp))
            if rc == -2:
                cap = int(n.value)
                continue
            assert rc == 0
            return out[:n.value].copy(), rl.value, mp[:nmp.value].copy()


# ---- batch calls -------------------------------------------------------------

def oracle_batch(par, off, a, n_segs=None, threads=4):
    tot = int(off[-1])
    f, p, v = (np.empty(max(tot, 1), np.int32) for _ in range(3))
    ns = None if n_segs is None else np.ascontiguousarray(n_segs, np.int32)
    ev = oracle().co_batch_fpv(C.byref(_co(par)), len(off) - 1, off.ctypes.data, a.ctypes.data,
                               None if ns is None else ns.ctypes.data, f.ctypes.data, p.ctypes.data, v.ctypes.data, threads)
    return f[:tot], p[:tot], v[:tot], ev


def wave_model_batch(par, off, a, n_segs=None, ring=128):
    tot = int(off[-1])
    f, p, v = (np.full(max(tot, 1), -77, np.int32) for _ in range(3))
    ns = None if n_segs is None else np.ascontiguousarray(n_segs, np.int32)
    stats = np.zeros(6, np.int64)
    oracle().wm_batch_fpv(C.byref(_co(par)), len(off) - 1, off.ctypes.data, a.ctypes.data,
                          None if ns is None else ns.ctypes.data, f.ctypes.data, p.ctypes.data, v.ctypes.data,
                          ring, stats.ctypes.data)
    names = ("lane_evals", "chunks", "deep_chunks", "general_walks", "units", "singletons")
    return f[:tot], p[:tot], v[:tot], dict(zip(names, stats.tolist()))


def time_top(par, off, a, threads, use_ref=False, n_segs=None, reps=1):
    """Wall seconds of the per-read top call over the batch (cpu_baseline leg)."""
    fn = None
    if use_ref:
        fn = C.cast(ref().mm_chain_dp_fpga, C.c_void_p)
    ns = None if n_segs is None else np.ascontiguousarray(n_segs, np.int32)
    chk = C.c_uint64(0)
    sec = oracle().co_time_top(C.byref(_co(par)), len(off) - 1, off.ctypes.data, a.ctypes.data,
                               None if ns is None else ns.ctypes.data, threads, reps, fn, C.byref(chk))
    return sec, chk.value
