"""DP parameter presets, as the reference derives them for mm_chain_dp_fpga.

Argument order and meaning follow reference chain.c:218 / mmpriv.h:69:
``max_dist_x`` = max_chain_gap_ref, ``max_dist_y`` = max_chain_gap_qry
(map.c:358-366, passed at map.c:525), ``bw``/``max_skip``/``min_sc``/``is_cdna``
from fpga_set_params (main.c:243), ``n_segs`` per read.
"""
import ctypes as C


class ChainParams(C.Structure):
    """Mirror of chaindp_params_t (include/chaindp.h) and co_params_t (oracle/chain_oracle.h)."""
    _fields_ = [(k, C.c_int32) for k in
                ("max_dist_x", "max_dist_y", "bw", "max_skip", "min_sc", "is_cdna", "n_segs")]

    def astuple(self):
        return tuple(getattr(self, k) for k, _ in self._fields_)

    def __repr__(self):
        return "ChainParams(" + ", ".join(f"{k}={getattr(self, k)}" for k, _ in self._fields_) + ")"


# options.c:29-34 (defaults), :84-87 (ava-ont), :88-92 (ava-pb), :95-96 (map-ont), :111-131 (sr), :132-139 (splice)
PRESETS = {
    "map-ont": dict(max_dist_x=5000, max_dist_y=5000, bw=500, max_skip=25, min_sc=40, is_cdna=0, n_segs=1),
    "ava-ont": dict(max_dist_x=10000, max_dist_y=10000, bw=500, max_skip=25, min_sc=100, is_cdna=0, n_segs=1),
    "ava-pb": dict(max_dist_x=10000, max_dist_y=10000, bw=2000, max_skip=25, min_sc=100, is_cdna=0, n_segs=1),
    # short paired reads, 2 x 150 bp: gap_ref = max(max_frag_len - qlen_sum, max_gap) = 500, gap_qry = max(qlen_sum, max_gap) = 300
    "sr": dict(max_dist_x=500, max_dist_y=300, bw=100, max_skip=25, min_sc=25, is_cdna=0, n_segs=2),
    "splice": dict(max_dist_x=200000, max_dist_y=2000, bw=200000, max_skip=25, min_sc=40, is_cdna=1, n_segs=1),
}


def preset(name, **overrides):
    d = dict(PRESETS[name])
    d.update(overrides)
    return ChainParams(**d)
