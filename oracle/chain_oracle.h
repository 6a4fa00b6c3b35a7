/*
 * chain_oracle.h -- CPU restatement of the reference's anchor-chaining DP.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (minimap2_chaindp_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked byte-for-byte
 * against the reference's own chain.c compiled unmodified in the build
 * container (oracle/Makefile -> oracle/_ref/), see tests/test_oracle_vs_ref.py,
 * and against the golden vectors that build emitted into tests/golden/.
 *
 * Reference (paths relative to the reference tree):
 *   chain.c:9-21     ilog2_32
 *   chain.c:218-327  mm_chain_dp_fpga  (device half: f/p/v recurrence + compaction)
 *   chain.c:329-431  mm_chain_dp_bottom (host half: backtrack)
 *   minimap.h:48-55  mm128_t, struct new_seed
 *   mmpriv.h:21-22   MM_SEED_SEG_SHIFT / MM_SEED_SEG_MASK
 *   ksort.h:101-151  radix_sort_128x / radix_sort_64 (unstable, order matters)
 *   hit.c:8-95       mm_cal_fuzzy_len, mm_reg_set_coor, hash64, mm_gen_regs (chains -> hits)
 *   esterr.c:7-64    get_for_qpos, get_mini_idx, mm_est_err (divergence estimate of a hit)
 */
#ifndef CHAIN_ORACLE_H
#define CHAIN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } co_anchor_t;              /* == mm128_t, minimap.h:48 */
typedef struct { co_anchor_t seed; int32_t p, f; } co_seed_t; /* == struct new_seed, minimap.h:51-55 (24 B) */

typedef struct {
	int32_t max_dist_x;  /* gap_ref   (first  argument of mm_chain_dp_fpga) */
	int32_t max_dist_y;  /* gap_qry   (second argument) */
	int32_t bw;
	int32_t max_skip;
	int32_t min_sc;
	int32_t is_cdna;
	int32_t n_segs;
} co_params_t;

/* chain.c:228-284: the score/backtrack recurrence.  f,p,v,t: caller-allocated
 * int32[n] (t is scratch and is zeroed here like chain.c:234).  Returns the
 * number of executions of the inner-loop body (chain.c:254), the "pair
 * evaluations" figure used by the benchmark. */
int64_t co_chain_fpv(const co_params_t *par, int64_t n, const co_anchor_t *a,
                     int32_t *f, int32_t *p, int32_t *v, int32_t *t);

/* chain.c:286-317: order-dependent compaction into new_seed[].  id is int32[n]
 * scratch, out has room for n records.  Returns new_i. */
uint32_t co_compact(const co_params_t *par, int64_t n, const co_anchor_t *a,
                    const int32_t *f, const int32_t *p, const int32_t *v,
                    co_seed_t *out, int32_t *id);

/* chain.c:218-327 as one call (does NOT free a, unlike the reference).
 * Returns a malloc'd co_seed_t[n] (caller frees) and *new_i. */
co_seed_t *co_chain_top(const co_params_t *par, int64_t n, const co_anchor_t *a, uint32_t *new_i);

/* chain.c:329-431.  Returns malloc'd anchors b[] grouped by chain (caller
 * frees), *n_u chains, *u (malloc'd, score<<32|count).  NULL when no chain. */
co_anchor_t *co_chain_bottom(int min_cnt, int min_sc, const co_seed_t *s, uint32_t new_i,
                             int *n_u, uint64_t **u);

/* ksort.h:101-151 restated (in-place MSD radix sort, 8-bit digits, insertion
 * sort below 65 elements). */
void co_radix_sort_128x(co_anchor_t *beg, co_anchor_t *end);
void co_radix_sort_64(uint64_t *beg, uint64_t *end);

/* == mm_reg1_t (minimap.h:100-115, 80 B) with two reserved words in the place of its mm_extra_t pointer;
 * `bits` is the bit-field word (rev = bit 10). */
typedef struct {
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0;
	uint32_t bits, hash;
	float div;
	uint32_t reserved[2];
} co_reg_t;                                                  /* 80 B */
#define CO_REG_REV_BIT 10

/* hit.c:52-95: the read's chains (u: score<<32|count, a: their anchors, chain after chain) as hits, best score
 * first; out has room for n_u records.  Fields the reference leaves zero stay zero; parent = -1, div = -1. */
void co_gen_regs(uint32_t hash, int32_t qlen, int32_t n_u, const uint64_t *u, const co_anchor_t *a, co_reg_t *out);

/* esterr.c:30-64: sets div of every hit from the minimizer positions of the read (mini_pos: q_span<<32 | pos) and
 * the reference lengths; n_match / n_tot (the integers div is made of) are returned too when the arrays are given. */
void co_est_err(const int32_t *ref_len, int32_t qlen, int32_t n_regs, co_reg_t *regs, const co_anchor_t *a,
                int32_t n, const uint64_t *mini_pos, int32_t *n_match, int32_t *n_tot);

/* Batch helpers used by tests and by bench.py's cpu_baseline leg.
 * Reads are given CSR-style: read r owns anchors [off[r], off[r+1]).
 * co_batch_fpv runs the recurrence for reads [r0,r1) on `threads` pthreads
 * (reads are dealt by cumulative anchor count); returns pair evaluations. */
int64_t co_batch_fpv(const co_params_t *par, int64_t n_reads, const int64_t *off,
                     const co_anchor_t *a, const int32_t *n_segs_per_read,
                     int32_t *f, int32_t *p, int32_t *v, int threads);

/* Timed variant: the full reference-shaped per-read call (malloc scratch,
 * recurrence, compaction, free) like mm_chain_dp_fpga, on `threads` pthreads,
 * the batch processed `reps` times; returns wall seconds between "all workers
 * ready" and "last worker done" (thread start-up and input copies excluded).
 * fn == NULL uses co_chain_top; otherwise fn must have the reference signature
 * (oracle/_ref's mm_chain_dp_fpga), which frees its input, so every worker
 * makes private malloc'd copies of its reads before the clock starts. */
typedef void *(*co_ref_top_fn)(int, int, int, int, int, int, int, int64_t, void *, uint32_t *);
double co_time_top(const co_params_t *par, int64_t n_reads, const int64_t *off,
                   const co_anchor_t *a, const int32_t *n_segs_per_read,
                   int threads, int reps, co_ref_top_fn fn, uint64_t *checksum);

#ifdef __cplusplus
}
#endif
#endif
