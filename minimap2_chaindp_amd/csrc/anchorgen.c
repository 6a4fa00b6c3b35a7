/*
 * anchorgen.c -- seeded generator of ONT-shaped anchor batches (host C, pthreads).
 *
 * The reference ships no read data and its BRANCH_MINIMAP2_DUMP_CHAINDP dump
 * branch is not in the tree (reference README.md:10), so the benchmark
 * workloads of BASELINE.json (configs 2-5) are synthesised in the layout the
 * chaining DP consumes: per read an array of mm128_t anchors (reference
 * minimap.h:48) sorted ascending by x, with
 *     x = strand<<63 | rid<<32 | ref_pos        (reference map.c:219-223, upstream layout)
 *     y = seg_id<<48 | q_span<<32 | query_pos   (reference map.c:220,224,227; mmpriv.h:21)
 * A read is a set of "hits" (one per overlapping target and strand): colinear
 * runs of anchors with geometric query steps and a slowly drifting diagonal
 * (indels), plus uniformly random noise anchors and optional exact-x ties.
 *
 * Generation is deterministic per (seed, read index) and independent of the
 * thread count.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

typedef struct { uint64_t x, y; } ag_anchor_t;

typedef struct {
	int32_t read_len;        /* mean read length (bp) */
	int32_t read_len_jitter; /* +- percent, uniform */
	int32_t n_hits;          /* mean hits (target x strand overlaps) per read */
	int32_t min_ovl_pct;     /* overlap length uniform in [min_ovl_pct, 100] % of the read */
	int32_t step;            /* mean query distance between consecutive anchors of a hit */
	int32_t indel_pct;       /* per-anchor probability (%) that the diagonal shifts */
	int32_t indel_max;       /* max shift per event (bp) */
	int32_t noise_pct;       /* random anchors, percent of the hit anchors */
	int32_t tie_pct;         /* percent of anchors duplicated with the same x (different query pos) */
	int32_t q_span;          /* k-mer span */
	int32_t span_jitter;     /* spans uniform in [q_span, q_span+span_jitter] */
	int32_t n_ref;           /* number of target sequences (rid range) */
	int32_t ref_len;         /* length of each target */
	int32_t n_segs;          /* >1: anchors get a random segment id in [0,n_segs) per hit */
	int32_t skew;            /* 1: per-read anchor budget log-uniform in [skew_min, skew_max] (config 5) */
	int32_t skew_min, skew_max;
} ag_config_t;

static inline uint64_t splitmix64(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
typedef struct { uint64_t s; } rng_t;
static inline uint64_t rnd(rng_t *r) /* xorshift64* */
{
	uint64_t x = r->s;
	x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
	r->s = x;
	return x * 0x2545F4914F6CDD1DULL;
}
static inline uint32_t rnd_below(rng_t *r, uint32_t n) { return n ? (uint32_t)((rnd(r) >> 32) * (uint64_t)n >> 32) : 0; }
static inline double rnd_unit(rng_t *r) { return (double)(rnd(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline int32_t rnd_geometric(rng_t *r, int32_t mean)
{
	double u;
	if (mean <= 1) return 1;
	u = rnd_unit(r);
	if (u < 1e-300) u = 1e-300;
	return 1 + (int32_t)(-log(u) * (double)(mean - 1));
}

typedef struct { ag_anchor_t *a; int64_t n, cap; } vec_t;
static inline void push(vec_t *v, uint64_t x, uint64_t y)
{
	if (v->n == v->cap) {
		v->cap = v->cap ? v->cap * 2 : 8192;
		v->a = (ag_anchor_t*)realloc(v->a, (size_t)v->cap * sizeof(ag_anchor_t));
	}
	v->a[v->n].x = x; v->a[v->n].y = y; ++v->n;
}

static int cmp_xy(const void *pa, const void *pb)
{
	const ag_anchor_t *a = (const ag_anchor_t*)pa, *b = (const ag_anchor_t*)pb;
	if (a->x != b->x) return a->x < b->x ? -1 : 1;
	if (a->y != b->y) return a->y < b->y ? -1 : 1;
	return 0;
}

static void gen_read(const ag_config_t *c, uint64_t seed, int64_t ridx, vec_t *v)
{
	uint64_t sm = seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(ridx + 1));
	rng_t r;
	int32_t L, hits, h, step = c->step > 0 ? c->step : 1;
	int64_t budget = -1, n_hit_anchors, k;
	r.s = splitmix64(&sm) | 1;
	v->n = 0;
	L = c->read_len;
	if (c->read_len_jitter > 0) {
		int32_t j = (int32_t)((int64_t)L * c->read_len_jitter / 100);
		L += (int32_t)rnd_below(&r, 2 * j + 1) - j;
	}
	if (L < 64) L = 64;
	hits = c->n_hits;
	if (c->skew) { /* log-uniform anchor budget; hits and read length follow the budget */
		double lo = log((double)(c->skew_min > 1 ? c->skew_min : 1)), hi = log((double)(c->skew_max > 1 ? c->skew_max : 1));
		budget = (int64_t)exp(lo + (hi - lo) * rnd_unit(&r));
		if (budget < 1) budget = 1;
		/* keep the per-hit shape (anchors per hit ~ L*0.55/step) and scale the number of hits, then the length */
		{
			double per_hit = (double)L * (100 + c->min_ovl_pct) / 200.0 / step;
			if (per_hit < 1) per_hit = 1;
			hits = (int32_t)((double)budget / per_hit + 0.5);
			if (hits < 1) { hits = 1; L = (int32_t)((double)budget * step * 200.0 / (100 + c->min_ovl_pct)); if (L < 64) L = 64; }
		}
	} else if (hits > 1) hits = hits / 2 + (int32_t)rnd_below(&r, (uint32_t)hits + 1); /* hits/2 .. 3hits/2 */
	for (h = 0; h < hits; ++h) {
		uint64_t rid = rnd_below(&r, (uint32_t)(c->n_ref > 0 ? c->n_ref : 1));
		uint64_t strand = rnd(&r) >> 63;
		uint64_t seg = c->n_segs > 1 ? rnd_below(&r, (uint32_t)c->n_segs) : 0;
		int32_t ovl = (int32_t)((int64_t)L * (c->min_ovl_pct + (int32_t)rnd_below(&r, (uint32_t)(101 - c->min_ovl_pct))) / 100);
		int32_t qs, q, drift = 0;
		int64_t rs;
		if (ovl < 1) ovl = 1;
		qs = (int32_t)rnd_below(&r, (uint32_t)(L - ovl + 1));
		rs = (int64_t)rnd_below(&r, (uint32_t)(c->ref_len > ovl + 1024 ? c->ref_len - ovl - 1024 : 1)) + 512;
		for (q = qs; q < qs + ovl; q += rnd_geometric(&r, step)) {
			int32_t span = c->q_span + (c->span_jitter > 0 ? (int32_t)rnd_below(&r, (uint32_t)c->span_jitter + 1) : 0);
			int64_t rpos;
			if (c->indel_pct > 0 && (int32_t)rnd_below(&r, 100) < c->indel_pct)
				drift += (int32_t)rnd_below(&r, 2 * (uint32_t)c->indel_max + 1) - c->indel_max;
			rpos = rs + (q - qs) + drift;
			if (rpos < 0) rpos = 0;
			push(v, strand << 63 | rid << 32 | (uint64_t)(uint32_t)rpos,
			     seg << 48 | (uint64_t)(span & 0xff) << 32 | (uint64_t)(uint32_t)(q + span - 1));
			if (c->tie_pct > 0 && (int32_t)rnd_below(&r, 100) < c->tie_pct) /* same x, another query position */
				push(v, strand << 63 | rid << 32 | (uint64_t)(uint32_t)rpos,
				     seg << 48 | (uint64_t)(span & 0xff) << 32 | (uint64_t)(uint32_t)(q + span - 1 + 1 + (int32_t)rnd_below(&r, 40)));
		}
	}
	n_hit_anchors = v->n;
	for (k = 0; k < n_hit_anchors * c->noise_pct / 100; ++k) {
		uint64_t rid = rnd_below(&r, (uint32_t)(c->n_ref > 0 ? c->n_ref : 1));
		uint64_t strand = rnd(&r) >> 63;
		uint64_t seg = c->n_segs > 1 ? rnd_below(&r, (uint32_t)c->n_segs) : 0;
		uint32_t rpos = rnd_below(&r, (uint32_t)(c->ref_len > 0 ? c->ref_len : 1));
		uint32_t q = rnd_below(&r, (uint32_t)L) + (uint32_t)c->q_span;
		push(v, strand << 63 | rid << 32 | rpos, seg << 48 | (uint64_t)(c->q_span & 0xff) << 32 | q);
	}
	qsort(v->a, (size_t)v->n, sizeof(ag_anchor_t), cmp_xy);
}

typedef struct {
	const ag_config_t *c;
	uint64_t seed;
	int64_t r0, r1, first_read;
	int64_t *counts;        /* pass 1: counts[r] */
	const int64_t *off;     /* pass 2 */
	ag_anchor_t *out;
} job_t;

static void *worker(void *arg)
{
	job_t *j = (job_t*)arg;
	vec_t v = {0, 0, 0};
	int64_t r;
	for (r = j->r0; r < j->r1; ++r) {
		gen_read(j->c, j->seed, j->first_read + r, &v);
		if (j->counts) j->counts[r] = v.n;
		else memcpy(j->out + j->off[r], v.a, (size_t)v.n * sizeof(ag_anchor_t));
	}
	free(v.a);
	return 0;
}

static void run(const ag_config_t *c, uint64_t seed, int64_t first_read, int64_t n_reads, int64_t *counts,
                const int64_t *off, ag_anchor_t *out, int threads)
{
	pthread_t *th;
	job_t *jobs;
	int k;
	if (threads < 1) threads = 1;
	if (threads > n_reads) threads = n_reads > 0 ? (int)n_reads : 1;
	th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
	jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
	for (k = 0; k < threads; ++k) {
		jobs[k].c = c; jobs[k].seed = seed; jobs[k].first_read = first_read;
		jobs[k].r0 = n_reads * k / threads; jobs[k].r1 = n_reads * (k + 1) / threads;
		jobs[k].counts = counts; jobs[k].off = off; jobs[k].out = out;
		pthread_create(&th[k], 0, worker, &jobs[k]);
	}
	for (k = 0; k < threads; ++k) pthread_join(th[k], 0);
	free(th); free(jobs);
}

/* Pass 1: off[0..n_reads] (CSR offsets, off[0]=0) for reads first_read .. first_read+n_reads-1.
 * Returns the total anchor count. */
int64_t ag_offsets(const ag_config_t *c, uint64_t seed, int64_t first_read, int64_t n_reads, int64_t *off, int threads)
{
	int64_t r, acc = 0;
	int64_t *counts = (int64_t*)malloc((size_t)(n_reads > 0 ? n_reads : 1) * 8);
	run(c, seed, first_read, n_reads, counts, 0, 0, threads);
	for (r = 0; r < n_reads; ++r) { off[r] = acc; acc += counts[r]; }
	off[n_reads] = acc;
	free(counts);
	return acc;
}

/* Pass 2: fill out[off[r] .. off[r+1]) for every read (same seed / first_read as pass 1). */
void ag_fill(const ag_config_t *c, uint64_t seed, int64_t first_read, int64_t n_reads, const int64_t *off,
             ag_anchor_t *out, int threads)
{
	run(c, seed, first_read, n_reads, 0, off, out, threads);
}
