/*
 * chaindp.h -- C ABI of the MI355X chaining-DP library (libchaindp_hip.so).
 *
 * This is the in-process entry to the device half of minimap2's anchor chaining
 * as the reference fork splits it:
 *
 *   reference chain.c:218-327  mm_chain_dp_fpga(max_dist_x, max_dist_y, bw, max_skip,
 *                              min_sc, is_cdna, n_segs, n, a, &new_i) -> new_seed[]
 *
 * Each chaindp_* call below processes a BATCH of reads (CSR layout) instead of one
 * read per call; per read the results are bit-identical to that function:
 * f[]/p[]/v[] are the arrays of chain.c:246-284, new_seed[] the compaction of
 * chain.c:286-317.  Plain pointers and sizes only; no torch or HIP types.
 * The packet-level driver ABI of the reference (fpga.h) is in chaindp_fpga.h.
 *
 * Types mirror the reference byte for byte:
 *   chaindp_anchor_t == mm128_t            (minimap.h:48)
 *   chaindp_seed_t   == struct new_seed    (minimap.h:51-55, 24 bytes)
 * Anchors of a read must be sorted ascending by x (map.c:233).
 */
#ifndef CHAINDP_H
#define CHAINDP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } chaindp_anchor_t;
typedef struct { chaindp_anchor_t seed; int32_t p, f; } chaindp_seed_t;

/* Arguments of mm_chain_dp_fpga in its order (chain.c:218); n_segs may be overridden
 * per read by the n_segs_per_read arrays below (collect_task_t::n_segs, fpga_chaindp.h:53). */
typedef struct {
	int32_t max_dist_x;   /* max_chain_gap_ref, map.c:361-366 */
	int32_t max_dist_y;   /* max_chain_gap_qry, map.c:358-360 */
	int32_t bw;
	int32_t max_skip;
	int32_t min_sc;
	int32_t is_cdna;
	int32_t n_segs;
} chaindp_params_t;

typedef struct chaindp_ctx chaindp_ctx_t;

#define CHAINDP_OK            0
#define CHAINDP_ERR_ARG      (-1)   /* bad argument (NULL pointer, negative size, negative gap/bw) */
#define CHAINDP_ERR_CAPACITY (-2)   /* batch larger than the context was created for */
#define CHAINDP_ERR_HIP      (-3)   /* a HIP runtime call failed; see chaindp_last_error() */
#define CHAINDP_ERR_NODEVICE (-4)   /* no usable GPU */

/* Number of HIP devices visible to the process (0 when there is none). */
int chaindp_device_count(void);

/* One context per (host thread, GPU): owns a stream and the device buffers for batches of up to
 * max_anchors anchors / max_reads reads (both at most 2^31-1; about 80 bytes of HBM per anchor).
 * Returns NULL on failure (no device, out of memory, bad arguments): chaindp_last_error(NULL) says why. */
chaindp_ctx_t *chaindp_create(int device, int64_t max_anchors, int64_t max_reads);
void chaindp_destroy(chaindp_ctx_t *ctx);
const char *chaindp_last_error(const chaindp_ctx_t *ctx);   /* ctx may be NULL: last create() error */

/* ---- host-buffer path (what the packet shim uses) -------------------------------------
 * off[n_reads+1]: CSR offsets into a[] (off[0] == 0); n_segs_per_read may be NULL.
 * chaindp_chain_batch = upload + run + download, synchronous.  v may be NULL. */
int chaindp_chain_batch(chaindp_ctx_t *ctx, const chaindp_params_t *par, int64_t n_reads, const int64_t *off,
                        const chaindp_anchor_t *a, const int32_t *n_segs_per_read,
                        int32_t *f, int32_t *p, int32_t *v);

/* The same in stages, so that a caller can keep a batch resident in HBM and time or
 * repeat the device stage alone.  upload/download are synchronous; run is asynchronous
 * on the context's stream until chaindp_sync().  (An anchor with nothing in reach on either side -- f = v = q_span, p = -1,
 * chain.c:251,283 with an empty window -- is only marked by the run; chaindp_download writes those entries before it copies, the
 * compaction works from the marks.  chaindp_run_device, which leaves the results in the caller's own arrays, writes them itself.) */
int chaindp_upload(chaindp_ctx_t *ctx, int64_t n_reads, const int64_t *off, const chaindp_anchor_t *a,
                   const int32_t *n_segs_per_read);
int chaindp_run(chaindp_ctx_t *ctx, const chaindp_params_t *par);
int chaindp_sync(chaindp_ctx_t *ctx);
int chaindp_download(chaindp_ctx_t *ctx, int32_t *f, int32_t *p, int32_t *v);

/* Compaction of the resident f/p/v into the wire format of the reference's result
 * packets (chain.c:286-317): seeds_off[n_reads+1] receives the CSR offsets of each
 * read's new_seed[] (seeds_off[r+1]-seeds_off[r] == new_i of read r), seeds[] the
 * records; seeds must have room for the batch's anchor count.  Requires a completed
 * chaindp_run on the same batch. */
int chaindp_compact(chaindp_ctx_t *ctx, const chaindp_params_t *par, int64_t *seeds_off, chaindp_seed_t *seeds);

/* chaindp_run followed by the compaction kernels, all asynchronous on the context's stream and
 * with every result left in HBM: the whole device half of the reference's split (chain.c:218-327)
 * for a resident batch.  This is what bench.py times. */
int chaindp_run_full(chaindp_ctx_t *ctx, const chaindp_params_t *par);

/* The HOST half of the reference's split as well, on the GPU: mm_chain_dp_bottom
 * (chain.c:329-431) for every read of the resident batch, from the compaction's new_seed[] (call after
 * chaindp_compact / chaindp_compact_offsets / chaindp_run_full).  min_sc comes from par, min_cnt is the
 * reference's first argument.  Outputs, per read r and in the reference's order (chains sorted by the x of their
 * first anchor, ties as its radix sort leaves them):
 *   chains_off[n_reads+1]   CSR offsets of the kept chains;  u[] = score<<32 | anchor count (the reference's u[])
 *   b_off[n_reads+1]        CSR offsets of the chained anchors;  b[] = the chains' anchors, one chain after the other
 * u needs room for chains_off[n_reads] entries and b for b_off[n_reads]; both are bounded by the batch's record
 * count (seeds_off[n_reads]).  Pass u = b = NULL to get only the offsets. */
int chaindp_backtrack(chaindp_ctx_t *ctx, const chaindp_params_t *par, int min_cnt,
                      int64_t *chains_off, uint64_t *u, int64_t *b_off, chaindp_anchor_t *b);

/* ---- chains to hits ----------------------------------------------------------------------
 * What the reference's host does with a read's chains before alignment: mm_gen_regs (hit.c:52-95, called at
 * map.c:862) and mm_est_err (esterr.c:30-64, called at map.c:872), over the chains the last chaindp_backtrack left
 * in HBM.  chaindp_reg_t is mm_reg1_t (minimap.h:100-115, 80 bytes) with two reserved words where its mm_extra_t
 * pointer is; `bits` is its bit-field word (mapq:8, split:2, rev:1, ... -- rev is bit 10). */
typedef struct {
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0;
	uint32_t bits, hash;
	float div;
	uint32_t reserved[2];
} chaindp_reg_t;

/* mm_gen_regs for every read of the batch: hash[r] and qlen[r] are the call's `hash` and `qlen` arguments for
 * read r; regs receives chains_off[n_reads] records (the offsets chaindp_backtrack returned), each read's hits in
 * the reference's order (by score and scrambled count, its unstable radix sort reproduced), `as` relative to the
 * read's chain anchors, parent = -1, div = -1, everything the reference leaves zero zero.  Bit-exact. */
int chaindp_gen_regs(chaindp_ctx_t *ctx, const uint32_t *hash, const int32_t *qlen, chaindp_reg_t *regs);

/* mm_est_err for hits the host hands back (after chain_post they may be fewer, split or reordered): read r owns
 * regs[regs_off[r] .. regs_off[r+1]); only cnt, as, rid, qs, rs, re and the rev bit are read, div is written.
 * ref_len[rid] is mi->seq[rid].len.  mini_pos_off / mini_pos: the reads' minimizer positions (q_span<<32 | pos,
 * map.c:141), or both NULL to use the ones chaindp_collect_seeds left resident.  match_tot, if given, receives
 * n_match and n_tot of esterr.c:53-61 for every hit (0, 0 where div stays -1).  n_match, n_tot and the two float
 * comparisons are exact; div = logf(n_tot / n_match) / avg_k uses the device's logf, within 2 ulp of the host's. */
int chaindp_est_err(chaindp_ctx_t *ctx, const int64_t *regs_off, chaindp_reg_t *regs, const int32_t *qlen,
                    const int32_t *ref_len, int32_t n_ref, const int64_t *mini_pos_off, const uint64_t *mini_pos,
                    int32_t *match_tot);

/* Scatter/gather variants used by the packet shim, whose reads sit in separate (pinned) packet
 * buffers: upload from one host pointer per read; run the compaction and return only the offsets;
 * then copy each read's new_seed[] straight to its place in a result packet (asynchronous on the
 * context's stream: call chaindp_sync() before touching dst). */
int chaindp_upload_gather(chaindp_ctx_t *ctx, int64_t n_reads, const int64_t *off,
                          const chaindp_anchor_t *const *read_anchors, const int32_t *n_segs_per_read);
int chaindp_compact_offsets(chaindp_ctx_t *ctx, const chaindp_params_t *par, int64_t *seeds_off);
int chaindp_download_seeds(chaindp_ctx_t *ctx, int64_t first_seed, int64_t n_seeds, chaindp_seed_t *dst);

/* The same two steps for callers whose per-read buffers are PINNED, device-visible host memory
 * (chaindp_host_alloc / hipHostMalloc): one kernel per direction streams every read between its host
 * buffer and HBM, with no per-read copy call and no host-side staging.  pinned = 1 in
 * chaindp_upload_gather_ex selects it (pinned = 0 behaves like chaindp_upload_gather).
 * chaindp_scatter_seeds writes read r's new_seed[] (after chaindp_compact_offsets) to dst[r] and zero-fills
 * up to the next 64-byte boundary, the padding of the reference's result packets (map.c:543-545); dst[r] ==
 * NULL skips the read.  dst[r] must be 16-byte aligned (the slots of a result packet are 64-byte aligned: the
 * device writes them in 16-byte stores over PCIe).  Asynchronous: chaindp_sync() before reading dst. */
int chaindp_upload_gather_ex(chaindp_ctx_t *ctx, int64_t n_reads, const int64_t *off,
                             const chaindp_anchor_t *const *read_anchors, const int32_t *n_segs_per_read, int pinned);
int chaindp_scatter_seeds(chaindp_ctx_t *ctx, int64_t n_reads, chaindp_seed_t *const *dst);

/* ---- seed collection on the GPU ----------------------------------------
 * What the reference's device does with a read's minimizers before chaining: collect_seed_hits (map.c:187-236,
 * called from fpga_work, map.c:523) -- lookup in the index image the host streams through fpga_load_index
 * (index.c:603-720: blobs B, H, V, P, types 4..7), skip_seed, the anchors in generation order, rep_len, mini_pos,
 * and radix_sort_128x exactly as the reference runs it (the order of equal x is input to the DP).
 * chaindp_index_create copies the four blobs to `device`.  chaindp_collect_seeds takes the minimizers of a batch
 * (CSR: mini_off[n_reads+1], mini as collect_minimizers leaves them, map.c:352; bid and qlen per read as in
 * collect_task_t) and leaves the sorted anchors resident in ctx exactly as chaindp_upload would (chaindp_run /
 * chaindp_run_full follow); it returns the per-read anchor offsets, rep_len and mini_pos offsets.  Synchronous. */
typedef struct chaindp_index chaindp_index_t;
chaindp_index_t *chaindp_index_create(int device, const void *B, size_t nB, const void *H, size_t nH,
                                      const void *V, size_t nV, const void *P, size_t nP);
void chaindp_index_destroy(chaindp_index_t *idx);
int chaindp_collect_seeds(chaindp_ctx_t *ctx, const chaindp_index_t *idx, int flag, int max_occ, int64_t n_reads,
                          const int64_t *mini_off, const chaindp_anchor_t *mini, const uint32_t *bid, const int32_t *qlen,
                          const int32_t *n_segs_per_read, int64_t *off, int32_t *rep_len, int64_t *mini_pos_off);
int chaindp_download_mini_pos(chaindp_ctx_t *ctx, uint64_t *mini_pos);      /* mini_pos_off[n_reads] entries */
/* The same with per-read buffers in PINNED host memory, moved by one kernel per direction (no staging copies):
 * read_mini[r] = read r's minimizers (mini_off gives the counts); dst[r] receives read r's mini_pos[] zero-padded
 * to 64 bytes (the reference's result packet layout, map.c:547-552), NULL skips the read; dst[r] 16-byte aligned.  The
 * scatter is asynchronous: chaindp_sync() before reading dst. */
int chaindp_collect_seeds_gather(chaindp_ctx_t *ctx, const chaindp_index_t *idx, int flag, int max_occ, int64_t n_reads,
                                 const int64_t *mini_off, const chaindp_anchor_t *const *read_mini, const uint32_t *bid,
                                 const int32_t *qlen, const int32_t *n_segs_per_read, int64_t *off, int32_t *rep_len,
                                 int64_t *mini_pos_off);
int chaindp_scatter_mini_pos(chaindp_ctx_t *ctx, int64_t n_reads, uint64_t *const *dst);
int chaindp_download_anchors(chaindp_ctx_t *ctx, chaindp_anchor_t *a);      /* the resident batch's anchors */

/* ---- the whole resident pipeline in one call: minimizers in, hits out (SURVEY 8f N2 -> 8a -> N1 -> N4) -----------
 * What the reference does per read between collect_minimizers and chain_post (map.c:350-366 on the host, 484-568 on the device,
 * 862 on the host again): collect_seed_hits over the index image, mm_chain_dp_fpga, mm_chain_dp_bottom, mm_gen_regs -- here
 * for a whole batch with nothing but the minimizers going in (16 B each, about 0.3 per anchor) and the hits coming out (80 B per
 * chain instead of 24 B per anchor): the anchors, f/p/v, new_seed[] and the chains never leave HBM.
 *   mini_off / mini / bid / qlen as in chaindp_collect_seeds; hash[r] as in chaindp_gen_regs;
 *   regs_off[n_reads + 1] receives the CSR offsets of every read's hits, regs the records (room for regs_cap of them:
 *   CHAINDP_ERR_CAPACITY, with regs_off filled in, if there are more -- call chaindp_gen_regs with a larger buffer then);
 *   rep_len (may be NULL) and n_anchors (may be NULL: the batch's seed count) are by-products.
 * Afterwards the batch is resident exactly as after the separate calls (chaindp_est_err, chaindp_download, ... work).
 * Bit-identical to the separate calls, hence to the reference (tests/test_gpu_regs.py). */
int chaindp_map_batch(chaindp_ctx_t *ctx, const chaindp_index_t *idx, int flag, int max_occ, const chaindp_params_t *par, int min_cnt,
                      int64_t n_reads, const int64_t *mini_off, const chaindp_anchor_t *mini, const uint32_t *bid, const int32_t *qlen,
                      const uint32_t *hash, int64_t *regs_off, chaindp_reg_t *regs, int64_t regs_cap, int32_t *rep_len, int64_t *n_anchors);

/* Pinned host memory (hipHostMalloc) for callers that want DMA-able staging buffers. */
void *chaindp_host_alloc(size_t bytes);
void chaindp_host_free(void *p);

/* ---- streaming pipeline ("H2D + kernels + D2H"; 8e: 2-3 streams per device) -------------------------
 * What the reference's driver did behind fpga_writebuf_submit / fpga_get_retbuf (fpga_chaindp.c:102-159,228-266):
 * batches stream through the device while the host keeps producing.  A pipe owns `depth` contexts (each with its own
 * stream and HBM buffers for max_anchors / max_reads) and `depth` pinned result buffers; chaindp_pipe_submit enqueues
 * H2D of the batch, prepass + chain DP + compaction on the next free slot and returns at once, so that with depth >= 3
 * the upload of batch n+1, the kernels of batch n and the download of batch n-1 overlap (the copy engines and the
 * shader array are independent).  Batches complete in submission order.
 *   off / a / n_segs_per_read as in chaindp_upload; they must stay valid until the batch has been waited for, and
 *   the copies are asynchronous only if they are pinned memory (chaindp_host_alloc) -- pageable memory is staged by
 *   the runtime, which serialises the pipe.
 *   chaindp_pipe_wait blocks for the OLDEST submitted batch, downloads exactly its new_seed[] records and
 *   describes them in *res (pinned, owned by the pipe, valid until chaindp_pipe_release, which frees the slot).
 *   Returns CHAINDP_ERR_BUSY from submit when every slot is in flight, from wait when nothing is. */
#define CHAINDP_ERR_BUSY     (-5)
typedef struct chaindp_pipe chaindp_pipe_t;
typedef struct {
	int64_t tag;                     /* the caller's tag of the batch */
	int64_t n_reads, n_anchors;      /* what was submitted */
	int64_t n_seeds;                 /* seeds_off[n_reads] */
	const int64_t *seeds_off;        /* [n_reads + 1] */
	const chaindp_seed_t *seeds;     /* [n_seeds], new_seed[] of every read, chain.c:286-317 */
} chaindp_pipe_result_t;
chaindp_pipe_t *chaindp_pipe_create(int device, int depth, int64_t max_anchors, int64_t max_reads);
void chaindp_pipe_destroy(chaindp_pipe_t *pipe);
int chaindp_pipe_submit(chaindp_pipe_t *pipe, const chaindp_params_t *par, int64_t n_reads, const int64_t *off,
                        const chaindp_anchor_t *a, const int32_t *n_segs_per_read, int64_t tag);
int chaindp_pipe_wait(chaindp_pipe_t *pipe, chaindp_pipe_result_t *res);
int chaindp_pipe_release(chaindp_pipe_t *pipe);
const char *chaindp_pipe_last_error(const chaindp_pipe_t *pipe);

/* ---- device-pointer path (inputs and outputs already in HBM) ---------------------------
 * All pointers are device addresses on ctx's GPU; stream is a hipStream_t passed as
 * void* (NULL = the context's own stream).  d_off int64[n_reads+1], d_a anchors,
 * d_n_segs int32[n_reads] or NULL, d_f/d_p/d_v int32[total_anchors] (d_v may NOT be NULL
 * here: the recurrence needs it as storage).  total_anchors must equal off[n_reads].
 * Asynchronous. */
int chaindp_run_device(chaindp_ctx_t *ctx, const chaindp_params_t *par, int64_t n_reads, int64_t total_anchors,
                       const void *d_off, const void *d_a, const void *d_n_segs,
                       void *d_f, void *d_p, void *d_v, void *stream);

/* ---- measurement ----------------------------------------------------------------------
 * With profiling on, every run brackets each kernel with HIP events on the stream it is
 * launched on; the accumulated device times are read back (after chaindp_sync) here.
 * ms[0] = prepass kernels, ms[1] = chain DP kernel, ms[2] = compaction kernels, ms[3] = backtrack kernels;
 * launches[i] = number of launches accumulated.  reset != 0 clears the accumulators. */
int chaindp_set_profiling(chaindp_ctx_t *ctx, int on);
int chaindp_get_kernel_ms(chaindp_ctx_t *ctx, double ms[4], int64_t launches[4], int reset);

/* Work decomposition of the last run: st[0] = units (independent DP problems, >= 2 anchors),
 * st[1] = singleton anchors resolved in the prepass, st[2] = anchors, st[3] = reads. */
int chaindp_get_stats(chaindp_ctx_t *ctx, int64_t st[4]);

/* Kernel tuning knob, mainly for tests: LDS ring capacity in anchors (128, 256 or 512). */
int chaindp_set_ring(chaindp_ctx_t *ctx, int ring);
/* force_general != 0 sends every unit through the general (64-bit, f64 gap cost) variant of the DP kernel
 * instead of the table-driven fast variant; results are identical, this exists for tests. */
int chaindp_set_variant(chaindp_ctx_t *ctx, int force_general);

#ifdef __cplusplus
}
#endif
#endif
