// chaindp_kernels.h -- internal interface between the C-ABI host code and the HIP kernels.
#ifndef CHAINDP_KERNELS_H
#define CHAINDP_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chaindp {

// arguments of mm_chain_dp_fpga (reference chain.c:218), same order as chaindp_params_t
struct Params {
	int32_t max_dist_x, max_dist_y, bw, max_skip, min_sc, is_cdna, n_segs;
};

// one independent DP problem: anchors [start, next gap > max_dist_x or end of read) of read `read`
struct Unit {
	int64_t start;   // global anchor index
	int32_t read;
	int32_t len;     // upper bound of its length (next unit's start or end of the read)
};

// What k_chain_twin needs of a unit besides its record, so that picking a unit up costs it no trips to sumq[] / off[] / the table:
// written by the prepass next to the unit list (same order).
struct UnitAux {
	int32_t rel0;    // the unit's first anchor, relative to its read
	uint32_t lutkey; // bits of the read's avg_qspan (chain.c:241): reads with equal keys have identical gap-cost tables
	uint32_t flags;  // bit 0: not for k_chain_twin (segment ids, a table that does not fit int8, a zero q_span)
	uint32_t pad;
};

// the anchor-parallel kernels of the prepass and of the compaction cut the batch into blocks of this many anchors
#define CHAINDP_BLOCK_ANCHORS 1024

// prepass scratch: one 64-bit unit-start mask per 64 anchors, per-block unit / singleton counts
struct PrepassScratch {
	uint64_t *start_mask;            // per 64 anchors: the unit starts among them
	uint64_t *single_mask;           // the singletons (anchors with an empty window whose successor starts anew: f = v = q_span, p = -1) ...
	uint64_t *emit_mask;             // ... and those of them that are emitted (q_span >= min_sc, chain.c:304)
	unsigned long long *block_cnt;   // per 1024-anchor block: units | singletons << 32; scanned in place
	unsigned long long *tile_tmp;    // scratch of the scan
	Unit *units_tmp;                 // units in anchor order, before the longest-first scatter
	unsigned int *hist;              // 2 x 128: length-class histogram / bases, cursors
	int2 *block_reads;               // per 1024-anchor block: reads of its first and last anchor (also used by the compaction)
	unsigned int *key_range;         // [0] smallest, [1] largest UnitAux::lutkey among the units k_chain_twin / k_chain_quad may take
};
size_t prepass_scratch_bytes(int64_t max_anchors, size_t *mask_bytes, size_t *blocks_bytes);

// counters[0] = units emitted (low 32 bits) | singleton anchors resolved by the prepass (high 32 bits)
hipError_t launch_prepass(hipStream_t st, const Params &par, int64_t n_reads, int64_t total, const int64_t *d_off, const void *d_a,
                          unsigned long long *d_sumq, Unit *d_units, unsigned long long *d_counters, PrepassScratch sc,
                          UnitAux *d_unit_aux = nullptr, const int32_t *d_n_segs = nullptr,
                          unsigned long long *d_left_cnt = nullptr);   // d_left_cnt: four hand-over words, zeroed with the batch's other accumulators
// f, p, v, flags[] of the batch's singletons (the prepass only marks them; the compaction reads the marks)
hipError_t launch_fill_singles(hipStream_t st, const Params &par, int64_t total, const void *d_a, PrepassScratch sc,
                               int32_t *d_f, int32_t *d_p, int32_t *d_v, uint8_t *d_flags);

// Per-read gap-cost table (uint16), lut_stride entries per read (multiple of 8); usable while
// bw <= CHAINDP_LUT_MAX_BW.  d_lut == nullptr makes every unit take the general (f64) variant.
#define CHAINDP_LUT_MAX_BW 4095
hipError_t launch_lut(hipStream_t st, const Params &par, int64_t n_reads, const int64_t *d_off,
                      unsigned long long *d_sumq, int lut_stride, uint16_t *d_lut);
size_t chain_lds_bytes(int ring, int lut_stride);

hipError_t launch_chain(hipStream_t st, int ring, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                        const int32_t *d_n_segs, const unsigned long long *d_sumq, const uint16_t *d_lut, int lut_stride,
                        const Unit *d_units, const unsigned long long *d_counters,
                        int32_t *d_f, int32_t *d_p, int32_t *d_v, unsigned long long *d_tg, uint32_t epoch, int32_t *d_first_child, uint8_t *d_flags,
                        const Unit *d_units_all = nullptr, const unsigned long long *d_counters_all = nullptr,
                        Unit *d_deep = nullptr, unsigned int *d_deep_cnt = nullptr, const unsigned int *d_long_units = nullptr,
                        int deep_eager = 0,    // tests: hand over any unit with a few deep scans
                        int deep_route = 0);   // 0: the batch decides which dense kernel runs; 1: k_chain_dense; 2: k_chain_dense1; 3: k_chain_dense16 (tests)
// *d_long_units: units of CHAINDP_LONG_UNIT anchors and more in the batch (PrepassScratch::hist + CHAINDP_LONG_UNIT_CLASS, valid
// after launch_prepass); above CHAINDP_DENSE_MAX_LONG of them nothing is handed over
#define CHAINDP_LONG_UNIT_CLASS 65      // hist[c] after k_unit_bases = units in length classes above c; class 65 ends at 8191 anchors
#define CHAINDP_DENSE_MAX_LONG 4096u
// Units whose scans keep reaching past the ring (dense repeats) are appended to d_deep / *d_deep_cnt by the launch above (when
// given) and redone by k_chain_dense (chaindp_dense.hip): a workgroup of four waves per unit, marks as one bit per distance in
// LDS.  The low 32 bits of *d_deep_cnt are the count.  Only units of at most CHAINDP_DENSE_BITCAP anchors are handed over (the
// bitmap covers that many distances), and only while 32-bit differences are exact over a ring of CHAINDP_DENSE_RING anchors.
#define CHAINDP_DENSE_BITCAP 65536
#define CHAINDP_DENSE_RING 512
#define CHAINDP_DENSE_UNITS 2048u      // units handed over per batch (about two rounds of workgroups on the chip)
hipError_t launch_chain_dense(hipStream_t st, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                              const uint16_t *d_lut, int lut_stride, const Unit *d_deep, const unsigned long long *d_deep_cnt,
                              int32_t *d_f, int32_t *d_p, int32_t *d_v, int32_t *d_first_child, uint8_t *d_flags,
                              const unsigned int *d_long_units, int deep_route, unsigned int *d_queue);   // d_queue: one zeroed word (the unit counter)
// the same with sixteen waves per unit (chaindp_dense.hip built with -DDN_VARIANT16): the device sends a short tail there
hipError_t launch_chain_dense16(hipStream_t st, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                              const uint16_t *d_lut, int lut_stride, const Unit *d_deep, const unsigned long long *d_deep_cnt,
                              int32_t *d_f, int32_t *d_p, int32_t *d_v, int32_t *d_first_child, uint8_t *d_flags,
                              const unsigned int *d_long_units, int deep_route, unsigned int *d_queue);   // d_queue: one zeroed word (the unit counter)
// ... or, when the batch is dense all over, by k_chain_dense1 (chaindp_dense1.hip): one wave per unit, many per CU, the same bit
// marks, deep chunks four at a time.  Both are launched; the device decides which one has work (dense_all(), chaindp_fast.h).
hipError_t launch_chain_dense1(hipStream_t st, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                               const uint16_t *d_lut, int lut_stride, const Unit *d_deep, const unsigned long long *d_deep_cnt,
                               const unsigned int *d_long_units, int deep_route, unsigned int *d_queues,   // d_queues: two zeroed words
                               int32_t *d_f, int32_t *d_p, int32_t *d_v, int32_t *d_first_child, uint8_t *d_flags);

// Two units per wave, 32 lanes each (chaindp_twin.hip): takes the ordinary units, appends the others (general-variant reads,
// scans that reach beyond 64 predecessors) to d_left / *d_left_cnt (low 32 bits = count), which launch_chain then runs.
hipError_t launch_chain_twin(hipStream_t st, const Params &par, int64_t max_units, const int64_t *d_off, const void *d_a,
                             const unsigned long long *d_sumq, const uint16_t *d_lut, int lut_stride, const Unit *d_units,
                             const unsigned long long *d_counters, int32_t *d_f, int32_t *d_p, int32_t *d_v,
                             int32_t *d_first_child, uint8_t *d_flags, Unit *d_left, unsigned int *d_left_cnt, int force_left, int64_t total,
                             const UnitAux *d_unit_aux, const unsigned int *d_route = nullptr, unsigned int *d_queue = nullptr   /* 8 x 64 words: the grab counters */);
size_t twin_lds_bytes();

// Four units per wave, 16 lanes each, two predecessors per lane (chaindp_quad.hip): takes a batch of ordinary units whose reads all
// have the same cost table (key_range: PrepassScratch::key_range) and says so in *d_route; otherwise it leaves the batch to
// launch_chain_twin, which is launched behind it and returns at once when *d_route is set.  Same hand-over list.
hipError_t launch_chain_quad(hipStream_t st, const Params &par, int64_t max_units, const void *d_a, const uint16_t *d_lut, int lut_stride,
                             const Unit *d_units, const UnitAux *d_unit_aux, const unsigned long long *d_counters, const unsigned int *d_key_range,
                             int32_t *d_f, int32_t *d_p, int32_t *d_v, int32_t *d_first_child, uint8_t *d_flags, Unit *d_left,
                             unsigned int *d_left_cnt, unsigned int *d_queue, unsigned int *d_route, int force_left, int64_t total);
size_t quad_lds_bytes();

// exclusive scan of n uint64 items in place (d_tile_tmp: ceil(n/1024)+1 words), total to *d_total
hipError_t launch_scan_u64(hipStream_t st, int64_t n, unsigned long long *d_data, unsigned long long *d_tile_tmp,
                           unsigned long long *d_total);

// compaction into new_seed[] (reference chain.c:286-317): see chaindp_compact.hip
struct CompactScratch {
	uint8_t *flags;                  // per anchor, written by the prepass / DP kernel: bit1 "emitted at its own step", bit2 "may be a first
	                                 // child", bits 3-4 the record flag bits (v >= min_sc, f < v).  (bit0 "late" lives in sub[].)
	unsigned long long *block_cnt;   // per 1024-anchor block record count; scanned in place
	unsigned long long *tile_tmp;
	unsigned long long *n_seeds;     // total records of the batch
	const int2 *block_reads;         // PrepassScratch::block_reads of the same batch
	const uint64_t *single_mask, *emit_mask;   // PrepassScratch's: flags[] of a singleton is whatever an earlier batch left there
	uint32_t *sub;                   // per run of 16 anchors (64 per 1024-anchor block; k_count): low half = records of the block in front
	                                 // of the run, high half = the late bits of its 16 anchors
};
size_t compact_scratch_bytes(int64_t max_anchors, size_t *flags_bytes, size_t *blocks_bytes);
hipError_t launch_compact(hipStream_t st, const Params &par, int64_t n_reads, int64_t total, const int64_t *d_off,
                          const void *d_a, const int32_t *d_f, const int32_t *d_p, const int32_t *d_v,
                          int32_t *d_first_child, int64_t *d_seeds_off, void *d_seeds,
                          CompactScratch sc);

// mm_chain_dp_bottom (reference chain.c:329-431) on the GPU: chaindp_bottom.hip.  m = records of the batch.
struct BottomScratch {
	uint8_t *has;                    // m
	int32_t *owner, *end_rec, *ccnt, *kpos, *bpos, *c_src, *c_dst;   // m each
	unsigned long long *key, *skey, *cu, *u_tmp, *u_out;             // m each
	void *b_tmp, *b_out, *w;         // m x 16 B each
	void *stacks;                    // (m/64 + 2 R + 4) x 12 B
	unsigned long long *block_cnt, *tile_tmp, *read_tot, *total;     // m/1024+1, same, R, 1
	int64_t *ends_off, *chains_off, *b_off;                          // R+1 each
};
hipError_t launch_backtrack(hipStream_t st, int min_cnt, int min_sc, int64_t n_reads, int64_t m_cap, const int64_t *d_soff, const void *d_seeds,
                            const unsigned long long *d_n_seeds, BottomScratch sc, int64_t n_seeds_host);

// chains to hits (mm_gen_regs, hit.c:52-95; mm_est_err, esterr.c:30-64): chaindp_regs.hip
hipError_t launch_gen_regs(hipStream_t st, int64_t n_reads, const int64_t *d_chains_off, const int64_t *d_b_off, const unsigned long long *d_u,
                           const void *d_b, const uint32_t *d_hash, const int32_t *d_qlen, void *d_z, void *d_stacks, void *d_regs);
hipError_t launch_est_err(hipStream_t st, int64_t n_reads, int64_t n_regs, const int64_t *d_regs_off, const int64_t *d_b_off, const void *d_b,
                          const int32_t *d_qlen, const int32_t *d_ref_len, int32_t n_ref, const int64_t *d_mp_off, const unsigned long long *d_mini_pos,
                          unsigned long long *d_sum_k, void *d_regs, int32_t *d_counts);

// seed collection on the GPU (reference map.c:112-236 over the FPGA index image, index.c:603-720): chaindp_seed.hip
// The four blobs of the image, as index.c:603-720 writes them:
//   B: per hash bucket 16 bytes: w0 = (p_off & 0xff) << 56 | n_buckets << 24;  w1 = h_off << 28 | p_off >> 8
//      (h_off in hash slots, rounded up to 8 per bucket; p_off in entries of P; an empty bucket is all zero)
//   H: per 8 hash slots 64 bytes: 4 B khash flag word (2 bits per slot, 16 slots), 8 x 6 B keys (low 48 bits), 12 B pad
//   V: per hash slot 8 bytes: the khash value (a position if the key's bit 0 is set, else p index << 32 | count)
//   P: 8 bytes per position
struct SeedIndex {                   // the four blobs in HBM
	const uint8_t *B, *H, *V, *P;
	uint64_t nB, nH, nV, nP;         // bytes
	int b_bits;                      // log2 of the bucket count
};
struct SeedScratch {                 // n = minimizers of the batch
	unsigned long long *kept, *used; // n each: anchors / used minimizers per minimizer, scanned in place
	unsigned long long *src, *mstate;// n each: where its hits are; hits | used << 32 | tandem << 33
	unsigned long long *tile_tmp;    // scan scratch, n / 1024 + 2
	unsigned long long *totals;      // 4: anchors, used minimizers, work items of the sort, units of the sort with equal x
	void *stacks;                    // (max_anchors / 64 + 2 R + 4) x 12 B: work items / range stacks of the per-read sort
	uint32_t *tied;                  // max_anchors / 64 + R + 8: the sort's units (reads, then work items) that hold equal x
};
// phase 1: probe, scans, per-read offsets and rep_len; the host then reads off[n_reads] (capacity check) and runs
// phase 2: expand (anchors in generation order into d_unsorted, mini_pos) and the per-read radix_sort_128x into d_a
hipError_t launch_seed_collect(hipStream_t st, const SeedIndex &ix, int flag, int max_occ, int64_t n_reads, int64_t n_mini,
                               const int64_t *d_mini_off, const void *d_mini, const uint32_t *d_bid, SeedScratch sc,
                               int64_t *d_off, int64_t *d_mp_off, int32_t *d_rep_len);
hipError_t launch_seed_expand_sort(hipStream_t st, const SeedIndex &ix, int flag, int64_t n_reads, int64_t n_mini,
                                   const int64_t *d_mini_off, const void *d_mini, const uint32_t *d_bid, const int32_t *d_qlen, SeedScratch sc,
                                   void *d_unsorted, void *d_a, const int64_t *d_off, unsigned long long *d_mini_pos, int max_n, int max_n2,
                                   int lab_cap, int64_t total);   // lab_cap: LDS bytes for digits in k_seed_sort_huge; total = d_off[n_reads]
// LDS bytes of the per-read sort for reads of up to max_n anchors in its first (workers = 32: sixteen waves with bucket
// tables, 1024 queue slots for small ranges) or second configuration (workers = 4: four waves, 256 slots); coop is unused.
// The host picks the largest max_n (<= 8192) and max_n2 that fit the device's LDS per workgroup.
size_t seed_sort_lds_bytes(int max_n, int workers, int coop);

// zero-copy movement between device-visible (pinned) host buffers and HBM: chaindp_io.hip
hipError_t launch_gather_reads(hipStream_t st, int64_t n_reads, const int64_t *d_off, const void *const *d_src, void *d_a);
hipError_t launch_scatter_seeds(hipStream_t st, int64_t n_reads, const int64_t *d_seeds_off, void *const *d_dst, const void *d_seeds);
// rounds bytes up to 16: both buffers must have that much room
hipError_t launch_copy_out(hipStream_t st, void *h_dst, const void *d_src, size_t bytes, int blocks);
hipError_t launch_scatter_words(hipStream_t st, int64_t n_reads, const int64_t *d_woff, void *const *d_dst, const void *d_words);

// The DP kernels address LDS by raw byte offsets from 0, so none of them may have static LDS in front of its dynamic segment.
// Checked once per kernel (the answer is a property of the code object), not once per launch.
hipError_t check_no_static_lds(const void *fn);

} // namespace chaindp
#endif
