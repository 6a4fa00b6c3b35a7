/*
 * chain_oracle.c -- CPU restatement of the reference's anchor-chaining DP.
 * TEST INFRASTRUCTURE ONLY (see chain_oracle.h for the rules and the parity
 * status: pinned against the compiled reference, oracle/_ref).
 *
 * Written from the behaviour of the reference, not from its text; each block
 * cites the reference lines whose arithmetic it must reproduce bit for bit.
 */
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include "chain_oracle.h"

#define SEG_SHIFT 48                      /* mmpriv.h:21 */
#define SEG_OF(y) ((int32_t)(((y) >> SEG_SHIFT) & 0xff)) /* mmpriv.h:22 */
#define SPAN_OF(y) ((int32_t)((y) >> 32 & 0xff))        /* chain.c:250 "only 8 bits of span" */

/* chain.c:9-21: floor(log2(v)) for v > 0 (table lookup there, bit scan here). */
static inline int floor_log2_u32(uint32_t v)
{
	return 31 - __builtin_clz(v);
}

/* chain.c:264-272: the gap cost of one (i,j) pair.  The only floating point on
 * the path: int -> double, two double multiplies, truncation. */
static inline int32_t pair_score(int32_t sc, int32_t dd, int64_t dr, int32_t dq,
                                 int same_seg, int is_cdna, float avg_qspan)
{
	int32_t log_dd = dd ? floor_log2_u32((uint32_t)dd) : 0;
	int32_t c_lin = (int)(dd * .01 * avg_qspan);
	if (is_cdna || !same_seg) {
		if (!same_seg && dr == 0) return sc + 1;                                /* chain.c:269 */
		if (dr > dq || !same_seg) return sc - (c_lin < log_dd ? c_lin : log_dd); /* chain.c:270 */
		return sc - (c_lin + (log_dd >> 1));                                    /* chain.c:271 */
	}
	return sc - (c_lin + (log_dd >> 1));                                        /* chain.c:272 */
}

int64_t co_chain_fpv(const co_params_t *par, int64_t n, const co_anchor_t *a,
                     int32_t *f, int32_t *p, int32_t *v, int32_t *t)
{
	const int max_dist_x = par->max_dist_x, max_dist_y = par->max_dist_y, bw = par->bw;
	const int max_skip = par->max_skip, is_cdna = par->is_cdna, n_segs = par->n_segs;
	int64_t i, j, st = 0, evals = 0;
	uint64_t sum_qspan = 0;
	float avg_qspan;

	if (n <= 0) return 0;
	memset(t, 0, (size_t)n * 4);                            /* chain.c:234 */
	for (i = 0; i < n; ++i) sum_qspan += (uint64_t)SPAN_OF(a[i].y); /* chain.c:240 */
	avg_qspan = (float)sum_qspan / n;                       /* chain.c:241 (f32 divide) */

	for (i = 0; i < n; ++i) {                               /* chain.c:246 */
		const uint64_t ri = a[i].x;
		const int32_t qi = (int32_t)a[i].y, q_span = SPAN_OF(a[i].y), sidi = SEG_OF(a[i].y);
		int32_t best = q_span, n_skip = 0;
		int64_t best_j = -1;
		/* chain.c:252: unsigned 64-bit compare against the int converted to u64 */
		while (st < i && ri - a[st].x > (uint64_t)(int64_t)max_dist_x) ++st;
		for (j = i - 1; j >= st; --j) {                     /* chain.c:253 */
			const int64_t dr = (int64_t)(ri - a[j].x);
			const int32_t dq = qi - (int32_t)a[j].y;
			const int same = sidi == SEG_OF(a[j].y);
			int32_t dd, sc;
			++evals;
			if ((same && dr == 0) || dq <= 0) continue;                  /* chain.c:257 */
			if ((same && dq > max_dist_y) || dq > max_dist_x) continue;  /* chain.c:258 */
			dd = (int32_t)(dr > dq ? dr - dq : dq - dr);                 /* chain.c:259 */
			if (same && dd > bw) continue;                               /* chain.c:260 */
			if (n_segs > 1 && !is_cdna && same && dr > max_dist_y) continue; /* chain.c:261 */
			sc = (int32_t)(dq < dr ? dq : dr);                           /* chain.c:262-263 */
			if (sc > q_span) sc = q_span;
			sc = pair_score(sc, dd, dr, dq, same, is_cdna, avg_qspan) + f[j]; /* chain.c:264-273 */
			if (sc > best) {                                             /* chain.c:274-276 */
				best = sc, best_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == i) {                                      /* chain.c:277-280 */
				if (++n_skip > max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = (int32_t)i;                         /* chain.c:281 */
		}
		f[i] = best, p[i] = (int32_t)best_j;                             /* chain.c:283 */
		v[i] = (best_j >= 0 && v[best_j] > best) ? v[best_j] : best;     /* chain.c:284 */
	}
	return evals;
}

uint32_t co_compact(const co_params_t *par, int64_t n, const co_anchor_t *a,
                    const int32_t *f, const int32_t *p, const int32_t *v,
                    co_seed_t *out, int32_t *id)
{
	const int min_sc = par->min_sc;
	uint32_t m = 0;
	int64_t i;
	for (i = 0; i < n; ++i) id[i] = -1;                     /* chain.c:236-237 */
	for (i = 0; i < n; ++i) {
		const int32_t pi = p[i];
		if (pi >= 0 && id[pi] == -1) {                      /* chain.c:287-303: late emission of a skipped predecessor */
			out[m].seed = a[pi];
			out[m].f = f[pi];
			out[m].p = (int32_t)((uint32_t)-1 << 2) | (v[pi] >= min_sc) | ((f[pi] < v[pi]) << 1);
			id[pi] = (int32_t)m++;
		}
		if (v[i] >= min_sc || pi >= 0) {                    /* chain.c:304-316 */
			out[m].seed = a[i];
			out[m].f = f[i];
			out[m].p = (pi >= 0 ? (int32_t)((uint32_t)id[pi] << 2) : (int32_t)((uint32_t)-1 << 2))
			         | (v[i] >= min_sc) | ((f[i] < v[i]) << 1);
			id[i] = (int32_t)m++;
		}
	}
	return m;
}

co_seed_t *co_chain_top(const co_params_t *par, int64_t n, const co_anchor_t *a, uint32_t *new_i)
{
	size_t nn = n > 0 ? (size_t)n : 1;
	int32_t *f = (int32_t*)malloc(nn * 4), *p = (int32_t*)malloc(nn * 4);
	int32_t *t = (int32_t*)malloc(nn * 4), *v = (int32_t*)malloc(nn * 4);
	int32_t *id = (int32_t*)malloc(nn * 4);
	co_seed_t *out = (co_seed_t*)malloc(nn * sizeof(co_seed_t));
	co_chain_fpv(par, n, a, f, p, v, t);
	*new_i = co_compact(par, n, a, f, p, v, out, id);
	free(f); free(p); free(t); free(v); free(id);
	return out;
}

/* ---- ksort.h:101-151 restated -------------------------------------------
 * The sort is unstable and the reference's output order for equal keys is a
 * property of this exact procedure (counting pass, cycle-leader permutation
 * walking buckets in ascending order, recursion on the next 8 bits, insertion
 * sort for <= 64 elements), so the oracle follows the same procedure.        */
#define RS_CUTOFF 64

#define DEFINE_RADIX(NAME, T, KEY)                                              \
static void NAME##_insertion(T *beg, T *end)                                    \
{                                                                               \
	T *i;                                                                       \
	for (i = beg + 1; i < end; ++i) {                                           \
		if (KEY(*i) < KEY(*(i - 1))) {                                          \
			T tmp = *i, *j = i;                                                 \
			while (j > beg && KEY(tmp) < KEY(*(j - 1))) { *j = *(j - 1); --j; } \
			*j = tmp;                                                           \
		}                                                                       \
	}                                                                           \
}                                                                               \
static void NAME##_msd(T *beg, T *end, int shift)                               \
{                                                                               \
	T *head[256], *tail[256];                                                   \
	size_t cnt[256];                                                            \
	int d;                                                                      \
	T *q;                                                                       \
	memset(cnt, 0, sizeof(cnt));                                                \
	for (q = beg; q != end; ++q) ++cnt[KEY(*q) >> shift & 0xff];                \
	for (d = 0, q = beg; d < 256; ++d) { head[d] = q; q += cnt[d]; tail[d] = q; } \
	for (d = 0; d < 256;) {                                                     \
		if (head[d] != tail[d]) {                                               \
			int l = (int)(KEY(*head[d]) >> shift & 0xff);                       \
			if (l != d) {                                                       \
				T carry = *head[d], swap;                                       \
				do {                                                            \
					swap = carry; carry = *head[l]; *head[l]++ = swap;          \
					l = (int)(KEY(carry) >> shift & 0xff);                      \
				} while (l != d);                                               \
				*head[d]++ = carry;                                             \
			} else ++head[d];                                                   \
		} else ++d;                                                             \
	}                                                                           \
	if (shift) {                                                                \
		int next = shift > 8 ? shift - 8 : 0;                                   \
		for (d = 0; d < 256; ++d) {                                             \
			T *b = tail[d] - cnt[d], *e = tail[d];                              \
			if (e - b > RS_CUTOFF) NAME##_msd(b, e, next);                      \
			else if (e - b > 1) NAME##_insertion(b, e);                         \
		}                                                                       \
	}                                                                           \
}                                                                               \
void NAME(T *beg, T *end)                                                       \
{                                                                               \
	if (end - beg <= RS_CUTOFF) NAME##_insertion(beg, end);                     \
	else NAME##_msd(beg, end, 56);                                              \
}

#define KEY_X(a) ((a).x)
#define KEY_ID(a) (a)
DEFINE_RADIX(co_radix_sort_128x, co_anchor_t, KEY_X)
DEFINE_RADIX(co_radix_sort_64, uint64_t, KEY_ID)

/* chain.c:329-431 */
co_anchor_t *co_chain_bottom(int min_cnt, int min_sc, const co_seed_t *s, uint32_t new_i,
                             int *n_u_out, uint64_t **u_out)
{
	const int64_t n = new_i;
	int32_t *order, *mark, n_u = 0, n_v = 0, k;
	int64_t i, j;
	uint64_t *u, *u2;
	co_anchor_t *b, *w, *tmp;

	*n_u_out = 0; *u_out = 0;
	order = (int32_t*)malloc((n ? n : 1) * 4);
	mark = (int32_t*)calloc(n ? n : 1, 4);
	/* chain.c:346-354: an element is a chain end if nobody points at it and its v>=min_sc flag is set */
	for (i = 0; i < n; ++i) if (s[i].p >= 0) mark[s[i].p >> 2] = 1;
	for (i = 0; i < n; ++i) if ((s[i].p & 1) && mark[i] == 0) ++n_u;
	if (n_u == 0) { free(order); free(mark); return 0; }
	u = (uint64_t*)malloc((size_t)n_u * 8);
	for (i = 0, n_u = 0; i < n; ++i) {
		if ((s[i].p & 1) && mark[i] == 0) {                 /* chain.c:362-370 */
			j = i;
			while (j >= 0 && (s[j].p & 2)) j = s[j].p >> 2; /* walk to the peak of f */
			if (j < 0) j = i;
			u[n_u++] = (uint64_t)(int64_t)s[j].f << 32 | (uint64_t)j;
		}
	}
	co_radix_sort_64(u, u + n_u);                           /* chain.c:371-375: ascending, then reversed */
	for (i = 0; i < n_u >> 1; ++i) { uint64_t x = u[i]; u[i] = u[n_u - i - 1]; u[n_u - i - 1] = x; }

	memset(mark, 0, (size_t)n * 4);                         /* chain.c:378-393: backtrack, best first */
	for (i = 0, k = 0; i < n_u; ++i) {
		const int32_t n_v0 = n_v, k0 = k;
		j = (int32_t)u[i];
		do {
			order[n_v++] = (int32_t)j;
			mark[j] = 1;
			j = s[j].p >> 2;
		} while (j >= 0 && mark[j] == 0);
		if (j < 0) {
			if (n_v - n_v0 >= min_cnt) u[k++] = u[i] >> 32 << 32 | (uint64_t)(n_v - n_v0);
		} else if ((int32_t)(u[i] >> 32) - s[j].f >= min_sc) {
			if (n_v - n_v0 >= min_cnt) u[k++] = ((u[i] >> 32) - (uint64_t)(int64_t)s[j].f) << 32 | (uint64_t)(n_v - n_v0);
		}
		if (k0 == k) n_v = n_v0;
	}
	n_u = k;
	free(mark);

	b = (co_anchor_t*)malloc((n_v ? n_v : 1) * sizeof(co_anchor_t)); /* chain.c:401-407 */
	for (i = 0, k = 0; i < n_u; ++i) {
		const int32_t k0 = k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) b[k++] = s[order[k0 + (ni - j - 1)]].seed;
	}
	free(order);

	w = (co_anchor_t*)malloc((n_u ? n_u : 1) * sizeof(co_anchor_t)); /* chain.c:412-426: chains re-ordered by first x */
	for (i = 0, k = 0; i < n_u; ++i) {
		w[i].x = b[k].x, w[i].y = (uint64_t)k << 32 | (uint64_t)i;
		k += (int32_t)u[i];
	}
	co_radix_sort_128x(w, w + n_u);
	tmp = (co_anchor_t*)malloc((n_v ? n_v : 1) * sizeof(co_anchor_t));
	u2 = (uint64_t*)malloc((n_u ? n_u : 1) * 8);
	for (i = 0, k = 0; i < n_u; ++i) {
		const int32_t src = (int32_t)w[i].y, cnt = (int32_t)u[src];
		u2[i] = u[src];
		memcpy(&tmp[k], &b[w[i].y >> 32], (size_t)cnt * sizeof(co_anchor_t));
		k += cnt;
	}
	memcpy(u, u2, (size_t)n_u * 8);
	memcpy(b, tmp, (size_t)k * sizeof(co_anchor_t));
	free(tmp); free(w); free(u2);
	*n_u_out = n_u; *u_out = u;
	return b;
}

/* ---- batch helpers (tests + cpu_baseline) -------------------------------- */

typedef struct {
	const co_params_t *par;
	const int64_t *off;
	const co_anchor_t *a;
	const int32_t *n_segs;
	int32_t *f, *p, *v;
	int64_t r0, r1, evals;
	co_ref_top_fn fn;
	int reps;
	pthread_barrier_t *start, *stop;
	uint64_t checksum;
} co_job_t;

static void split_by_anchors(int64_t n_reads, const int64_t *off, int threads, int64_t *cut)
{
	int64_t total = off[n_reads] - off[0], r = 0;
	int k;
	cut[0] = 0;
	for (k = 1; k < threads; ++k) {
		int64_t want = off[0] + total * k / threads;
		while (r < n_reads && off[r] < want) ++r;
		cut[k] = r;
	}
	cut[threads] = n_reads;
}

static void *fpv_worker(void *arg)
{
	co_job_t *jb = (co_job_t*)arg;
	int64_t r, cap = 0;
	int32_t *t = 0;
	for (r = jb->r0; r < jb->r1; ++r) {
		int64_t o = jb->off[r], n = jb->off[r + 1] - o;
		co_params_t par = *jb->par;
		if (jb->n_segs) par.n_segs = jb->n_segs[r];
		if (n > cap) { free(t); cap = n; t = (int32_t*)malloc((size_t)cap * 4); }
		jb->evals += co_chain_fpv(&par, n, jb->a + o, jb->f + o, jb->p + o, jb->v + o, t);
	}
	free(t);
	return 0;
}

int64_t co_batch_fpv(const co_params_t *par, int64_t n_reads, const int64_t *off,
                     const co_anchor_t *a, const int32_t *n_segs_per_read,
                     int32_t *f, int32_t *p, int32_t *v, int threads)
{
	pthread_t *th;
	co_job_t *jobs;
	int64_t *cut, evals = 0;
	int k;
	if (threads < 1) threads = 1;
	th = (pthread_t*)malloc(threads * sizeof(pthread_t));
	jobs = (co_job_t*)calloc(threads, sizeof(co_job_t));
	cut = (int64_t*)malloc((threads + 1) * 8);
	split_by_anchors(n_reads, off, threads, cut);
	for (k = 0; k < threads; ++k) {
		jobs[k].par = par, jobs[k].off = off, jobs[k].a = a, jobs[k].n_segs = n_segs_per_read;
		jobs[k].f = f, jobs[k].p = p, jobs[k].v = v, jobs[k].r0 = cut[k], jobs[k].r1 = cut[k + 1];
		pthread_create(&th[k], 0, fpv_worker, &jobs[k]);
	}
	for (k = 0; k < threads; ++k) { pthread_join(th[k], 0); evals += jobs[k].evals; }
	free(th); free(jobs); free(cut);
	return evals;
}

static void *top_worker(void *arg)
{
	co_job_t *jb = (co_job_t*)arg;
	int64_t r;
	uint64_t h = 0;
	int rep;
	void ***copies = 0;
	if (jb->fn) { /* the reference frees its input (chain.c:322): private malloc'd copies, made before the clock starts */
		copies = (void***)malloc((size_t)jb->reps * sizeof(void**));
		for (rep = 0; rep < jb->reps; ++rep) {
			copies[rep] = (void**)malloc((size_t)(jb->r1 - jb->r0 + 1) * sizeof(void*));
			for (r = jb->r0; r < jb->r1; ++r) {
				size_t bytes = (size_t)(jb->off[r + 1] - jb->off[r]) * sizeof(co_anchor_t);
				copies[rep][r - jb->r0] = malloc(bytes ? bytes : 1);
				memcpy(copies[rep][r - jb->r0], jb->a + jb->off[r], bytes);
			}
		}
	}
	pthread_barrier_wait(jb->start);
	for (rep = 0; rep < jb->reps; ++rep) {
		for (r = jb->r0; r < jb->r1; ++r) {
			int64_t o = jb->off[r], n = jb->off[r + 1] - o;
			co_params_t par = *jb->par;
			uint32_t new_i = 0, k;
			co_seed_t *s;
			if (jb->n_segs) par.n_segs = jb->n_segs[r];
			if (jb->fn) s = (co_seed_t*)jb->fn(par.max_dist_x, par.max_dist_y, par.bw, par.max_skip, par.min_sc,
			                                   par.is_cdna, par.n_segs, n, copies[rep][r - jb->r0], &new_i);
			else s = co_chain_top(&par, n, jb->a + o, &new_i);
			if (rep == 0) for (k = 0; k < new_i; ++k) h = h * 1099511628211ULL + (uint64_t)(uint32_t)s[k].f * 31u + (uint32_t)s[k].p;
			free(s);
		}
	}
	pthread_barrier_wait(jb->stop);
	if (copies) { for (rep = 0; rep < jb->reps; ++rep) free(copies[rep]); free(copies); }
	jb->checksum = h;
	return 0;
}

double co_time_top(const co_params_t *par, int64_t n_reads, const int64_t *off,
                   const co_anchor_t *a, const int32_t *n_segs_per_read,
                   int threads, int reps, co_ref_top_fn fn, uint64_t *checksum)
{
	pthread_t *th;
	co_job_t *jobs;
	int64_t *cut;
	pthread_barrier_t start, stop;
	struct timespec t0, t1;
	uint64_t h = 0;
	int k;
	if (threads < 1) threads = 1;
	if (reps < 1) reps = 1;
	th = (pthread_t*)malloc(threads * sizeof(pthread_t));
	jobs = (co_job_t*)calloc(threads, sizeof(co_job_t));
	cut = (int64_t*)malloc((threads + 1) * 8);
	split_by_anchors(n_reads, off, threads, cut);
	pthread_barrier_init(&start, 0, (unsigned)threads + 1);
	pthread_barrier_init(&stop, 0, (unsigned)threads + 1);
	for (k = 0; k < threads; ++k) {
		jobs[k].par = par, jobs[k].off = off, jobs[k].a = a, jobs[k].n_segs = n_segs_per_read;
		jobs[k].r0 = cut[k], jobs[k].r1 = cut[k + 1], jobs[k].fn = fn, jobs[k].reps = reps;
		jobs[k].start = &start, jobs[k].stop = &stop;
		pthread_create(&th[k], 0, top_worker, &jobs[k]);
	}
	pthread_barrier_wait(&start);          /* every worker is up and has its private input copies */
	clock_gettime(CLOCK_MONOTONIC, &t0);
	pthread_barrier_wait(&stop);           /* the slowest worker has finished its last read */
	clock_gettime(CLOCK_MONOTONIC, &t1);
	for (k = 0; k < threads; ++k) { pthread_join(th[k], 0); h ^= jobs[k].checksum + (uint64_t)k; }
	pthread_barrier_destroy(&start); pthread_barrier_destroy(&stop);
	if (checksum) *checksum = h;
	free(th); free(jobs); free(cut);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- hit.c:8-95 ------------------------------------------------------------ */

static uint64_t co_hash64(uint64_t key)                      /* hit.c:40-50 */
{
	key = (~key + (key << 21));
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4));
	key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}

static void co_reg_set_coor(co_reg_t *r, int32_t qlen, const co_anchor_t *a)   /* hit.c:8-38 */
{
	const int32_t k = r->as, last = r->as + r->cnt - 1, q_span = (int32_t)(a[k].y >> 32 & 0xff);
	int32_t i;
	const int rev = (int)(a[k].x >> 63);
	r->bits = (r->bits & ~(1u << CO_REG_REV_BIT)) | (uint32_t)rev << CO_REG_REV_BIT;
	r->rid = (int32_t)(a[k].x << 1 >> 33);
	r->rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
	r->re = (int32_t)a[last].x + 1;
	if (!rev) {
		r->qs = (int32_t)a[k].y + 1 - q_span;
		r->qe = (int32_t)a[last].y + 1;
	} else {
		r->qs = qlen - ((int32_t)a[last].y + 1);
		r->qe = qlen - ((int32_t)a[k].y + 1 - q_span);
	}
	r->mlen = r->blen = 0;
	if (r->cnt <= 0) return;
	r->mlen = r->blen = q_span;
	for (i = k + 1; i <= last; ++i) {
		const int32_t span = (int32_t)(a[i].y >> 32 & 0xff);
		const int32_t tl = (int32_t)a[i].x - (int32_t)a[i - 1].x;
		const int32_t ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
		r->blen += tl > ql ? tl : ql;
		r->mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
	}
}

void co_gen_regs(uint32_t hash, int32_t qlen, int32_t n_u, const uint64_t *u, const co_anchor_t *a, co_reg_t *out)
{
	co_anchor_t *z, tmp;
	int32_t i, k;
	if (n_u <= 0) return;
	z = (co_anchor_t*)malloc((size_t)n_u * sizeof(co_anchor_t));
	for (i = k = 0; i < n_u; ++i) {                          /* hit.c:61-68: key = score and count, low bits scrambled */
		const uint32_t h = (uint32_t)co_hash64((co_hash64(a[k].x) + co_hash64(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)u[i];
		k += (int32_t)u[i];
	}
	co_radix_sort_128x(z, z + n_u);
	for (i = 0; i < n_u >> 1; ++i) tmp = z[i], z[i] = z[n_u - 1 - i], z[n_u - 1 - i] = tmp;   /* larger score first */
	memset(out, 0, (size_t)n_u * sizeof(co_reg_t));
	for (i = 0; i < n_u; ++i) {                              /* hit.c:74-86 */
		co_reg_t *r = &out[i];
		r->id = i;
		r->parent = -1;                                      /* MM_PARENT_UNSET, mmpriv.h */
		r->score = r->score0 = (int32_t)(z[i].x >> 32);
		r->hash = (uint32_t)z[i].x;
		r->cnt = (int32_t)z[i].y;
		r->as = (int32_t)(z[i].y >> 32);
		r->div = -1.0f;
		co_reg_set_coor(r, qlen, a);
	}
	free(z);
}

/* ---- esterr.c:7-64 ---------------------------------------------------------- */

static int32_t co_for_qpos(int32_t qlen, const co_anchor_t *a)      /* esterr.c:7-14 */
{
	int32_t x = (int32_t)a->y;
	const int32_t q_span = (int32_t)(a->y >> 32 & 0xff);
	if (a->x >> 63) x = qlen - 1 - (x + 1 - q_span);
	return x;
}

void co_est_err(const int32_t *ref_len, int32_t qlen, int32_t n_regs, co_reg_t *regs, const co_anchor_t *a,
                int32_t n, const uint64_t *mini_pos, int32_t *n_match_out, int32_t *n_tot_out)
{
	int32_t i;
	uint64_t sum_k = 0;
	float avg_k;
	if (n == 0) return;
	for (i = 0; i < n; ++i) sum_k += mini_pos[i] >> 32 & 0xff;
	avg_k = (float)sum_k / n;
	for (i = 0; i < n_regs; ++i) {
		co_reg_t *r = &regs[i];
		const int rev = (int)(r->bits >> CO_REG_REV_BIT & 1);
		int32_t st, en, j, k, n_match, n_tot, x, L = 0, R = n - 1;
		r->div = -1.0f;
		if (n_match_out) n_match_out[i] = 0;
		if (n_tot_out) n_tot_out[i] = 0;
		if (r->cnt == 0) continue;
		x = co_for_qpos(qlen, rev ? &a[r->as + r->cnt - 1] : &a[r->as]);
		st = -1;
		while (L <= R) {                                     /* esterr.c:16-28 */
			const int32_t m = (int32_t)(((uint64_t)L + R) >> 1), y = (int32_t)mini_pos[m];
			if (y < x) L = m + 1;
			else if (y > x) R = m - 1;
			else { st = m; break; }
		}
		if (st < 0) continue;
		en = st;
		for (k = 1, j = st + 1, n_match = 1; j < n && k < r->cnt; ++j) {
			x = co_for_qpos(qlen, rev ? &a[r->as + r->cnt - 1 - k] : &a[r->as + k]);
			if (x == (int32_t)mini_pos[j]) ++k, en = j, ++n_match;
		}
		n_tot = en - st + 1;
		if (r->qs > avg_k && r->rs > avg_k) ++n_tot;
		if (qlen - r->qs > avg_k && ref_len[r->rid] - r->re > avg_k) ++n_tot;
		r->div = logf((float)n_tot / n_match) / avg_k;
		if (n_match_out) n_match_out[i] = n_match;
		if (n_tot_out) n_tot_out[i] = n_tot;
	}
}
