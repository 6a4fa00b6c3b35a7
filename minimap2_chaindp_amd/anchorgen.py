"""ctypes front end of csrc/anchorgen.c: seeded ONT-shaped anchor batches.

A batch is CSR-shaped: ``off`` (int64[n_reads+1]) and ``anchors``
(uint64[total, 2], column 0 = x, column 1 = y; byte-identical to an array of the
reference's ``mm128_t``, minimap.h:48).  Shapes follow BASELINE.json's configs;
the presets below are the ones bench.py and the tests name.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "csrc", "libanchorgen.so")


class AgConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "read_len", "read_len_jitter", "n_hits", "min_ovl_pct", "step", "indel_pct", "indel_max",
        "noise_pct", "tie_pct", "q_span", "span_jitter", "n_ref", "ref_len", "n_segs",
        "skew", "skew_min", "skew_max")]


# name -> (generator shape, DP parameter preset name)
PRESETS = {
    # ava-ont self-overlap, 10 kb reads (BASELINE configs 2 and 4): ~40 partial overlaps per read,
    # both strands, ~5.5k anchors per read.
    "ava-ont": dict(read_len=10000, read_len_jitter=0, n_hits=40, min_ovl_pct=10, step=40, indel_pct=30,
                    indel_max=6, noise_pct=8, tie_pct=1, q_span=15, span_jitter=0, n_ref=100000,
                    ref_len=10000 + 2048, n_segs=1, skew=0, skew_min=0, skew_max=0),
    # map-ont against a human-size reference (config 3): 1-3 long colinear runs per read.
    "map-ont": dict(read_len=8000, read_len_jitter=60, n_hits=2, min_ovl_pct=60, step=3, indel_pct=25,
                    indel_max=8, noise_pct=6, tie_pct=1, q_span=15, span_jitter=0, n_ref=24,
                    ref_len=120_000_000, n_segs=1, skew=0, skew_min=0, skew_max=0),
    # config 5: skewed batch, 1e2..1e5 anchors per read (log-uniform).
    "skew": dict(read_len=10000, read_len_jitter=0, n_hits=40, min_ovl_pct=10, step=40, indel_pct=30,
                 indel_max=6, noise_pct=8, tie_pct=1, q_span=15, span_jitter=0, n_ref=100000,
                 ref_len=10000 + 2048, n_segs=1, skew=1, skew_min=100, skew_max=100000),
    # adversarial shapes for the parity tests
    "ties": dict(read_len=3000, read_len_jitter=20, n_hits=6, min_ovl_pct=30, step=6, indel_pct=50,
                 indel_max=20, noise_pct=15, tie_pct=25, q_span=12, span_jitter=9, n_ref=3,
                 ref_len=20000, n_segs=1, skew=0, skew_min=0, skew_max=0),
    "paired": dict(read_len=300, read_len_jitter=0, n_hits=4, min_ovl_pct=50, step=4, indel_pct=10,
                   indel_max=3, noise_pct=10, tie_pct=5, q_span=15, span_jitter=6, n_ref=2,
                   ref_len=5000, n_segs=2, skew=0, skew_min=0, skew_max=0),
    # dense repeats: huge windows, few marks -> exercises the deep (beyond-LDS) path
    "dense": dict(read_len=6000, read_len_jitter=0, n_hits=30, min_ovl_pct=80, step=5, indel_pct=60,
                  indel_max=40, noise_pct=30, tie_pct=10, q_span=15, span_jitter=0, n_ref=1,
                  ref_len=9000, n_segs=1, skew=0, skew_min=0, skew_max=0),
}

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run __graft_entry__.build() first")
        lib = C.CDLL(_LIB_PATH)
        lib.ag_offsets.restype = C.c_int64
        lib.ag_offsets.argtypes = [C.POINTER(AgConfig), C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_int]
        lib.ag_fill.restype = None
        lib.ag_fill.argtypes = [C.POINTER(AgConfig), C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        _lib = lib
    return _lib


def make_config(preset="ava-ont", **overrides):
    d = dict(PRESETS[preset])
    d.update(overrides)
    return AgConfig(**d)


def offsets(preset="ava-ont", n_reads=1000, seed=1, first_read=0, threads=None, **overrides):
    """CSR offsets (int64[n_reads+1]) of reads first_read .. first_read+n_reads-1 without their anchors: what a dealer needs to
    split a job by anchor count."""
    lib = _load()
    cfg = make_config(preset, **overrides)
    threads = threads or min(32, os.cpu_count() or 1)
    off = np.zeros(n_reads + 1, dtype=np.int64)
    lib.ag_offsets(C.byref(cfg), seed, first_read, n_reads, off.ctypes.data, threads)
    return off


def generate(preset="ava-ont", n_reads=1000, seed=1, first_read=0, threads=None, out=None, **overrides):
    """Returns (off int64[n_reads+1], anchors uint64[total,2]).

    ``out`` may be a preallocated uint64[>=total,2] array (e.g. a view of pinned memory)."""
    lib = _load()
    cfg = make_config(preset, **overrides)
    threads = threads or min(32, os.cpu_count() or 1)
    off = np.zeros(n_reads + 1, dtype=np.int64)
    total = lib.ag_offsets(C.byref(cfg), seed, first_read, n_reads, off.ctypes.data, threads)
    if out is None:
        out = np.empty((total, 2), dtype=np.uint64)
    assert out.dtype == np.uint64 and out.shape[0] >= total and out.flags.c_contiguous
    lib.ag_fill(C.byref(cfg), seed, first_read, n_reads, off.ctypes.data, out.ctypes.data, threads)
    return off, out[:total]
