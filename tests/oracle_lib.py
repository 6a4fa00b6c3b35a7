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

SEED_DTYPE = np.dtype([("x", "<u8"), ("y", "<u8"), ("p", "<i4"), ("f", "<i4")])  # struct new_seed, 24 B
assert SEED_DTYPE.itemsize == 24


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
