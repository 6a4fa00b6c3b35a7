/*
 * wave_model.c -- CPU model of the GPU *formulation* of the chaining DP.
 * TEST INFRASTRUCTURE ONLY (same rules as chain_oracle.h).
 *
 * The HIP kernels (minimap2_chaindp_amd/csrc/chaindp_kernels.hip) do not run
 * the reference's scalar loop; they run a 64-lane chunked re-formulation of
 * it.  This file states that re-formulation in plain C, lane by lane, so that
 * its equivalence to the scalar recurrence (reference chain.c:246-284,
 * restated in chain_oracle.c) can be checked on the CPU, without a GPU, on
 * millions of anchors.  It is a model of the algorithm, not of the hardware.
 *
 * The four derivations being checked (DESIGN.md "Wave formulation"):
 *  1. units: anchors split where a[i].x - a[i-1].x > max_dist_x are independent
 *     DP problems (given the read-level avg_qspan); singletons have f=v=span, p=-1.
 *  2. a chunk of 64 predecessors j = i-1-64c-k (lane k) can be evaluated at once:
 *     "new max" lanes are those whose score beats the exclusive prefix max
 *     (seeded with the running max), ties keep the larger j.
 *  3. t[] marks may be written for all filter-passing lanes before any lane reads
 *     its own mark (a mark on j only ever comes from a larger j, i.e. a lower lane).
 *  4. n_skip is a walk clamped at 0: with S_k = n0 + #B(<=k) - #A(<=k),
 *     n_skip_k = S_k - min(0, min_{m<=k} S_m); the first B lane with n_skip_k >
 *     max_skip is the break; lanes beyond it contribute nothing.
 * plus the storage split the kernel uses: a ring of the last RING anchors (LDS
 * in the kernel) and "deep" accesses to the full arrays (global memory), with t[]
 * marks routed to whichever side holds the target.
 */
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "chain_oracle.h"

#define W 64
#define SEG_OF(y) ((int32_t)(((y) >> 48) & 0xff))
#define SPAN_OF(y) ((int32_t)((y) >> 32 & 0xff))

typedef struct {
	int64_t lane_evals;   /* lanes evaluated, including those wasted past a break */
	int64_t chunks, deep_chunks, general_walks, units, singletons;
} wm_stats_t;

static int32_t gap_cost_score(int32_t sc, int32_t dd, int64_t dr, int32_t dq, int same, int is_cdna, float avg)
{
	int32_t lg = dd ? 31 - __builtin_clz((uint32_t)dd) : 0;
	int32_t lin = (int)(dd * .01 * avg);
	if (is_cdna || !same) {
		if (!same && dr == 0) return sc + 1;
		if (dr > dq || !same) return sc - (lin < lg ? lin : lg);
		return sc - (lin + (lg >> 1));
	}
	return sc - (lin + (lg >> 1));
}

/* One unit [u0,u1) of one read.  ring = modelled LDS capacity in anchors (multiple of 64).
 * tg[] is the "global" mark array (zeroed by the caller once per batch), tl[] the ring's. */
static void wm_unit(const co_params_t *par, float avg, const co_anchor_t *a, int64_t u0, int64_t u1,
                    int32_t *f, int32_t *p, int32_t *v, int32_t *tg, int ring, wm_stats_t *st)
{
	const uint64_t maxx = (uint64_t)(int64_t)par->max_dist_x;
	int32_t *tl = (int32_t*)calloc(ring, 4);
	int64_t i;
	for (i = u0; i < u1; ++i) {
		const uint64_t ri = a[i].x;
		const int32_t qi = (int32_t)a[i].y, span = SPAN_OF(a[i].y), sidi = SEG_OF(a[i].y);
		const int32_t tag = (int32_t)(i - u0) + 1;   /* any value unique per i within the unit, never 0 */
		int32_t max_f = span, n_skip = 0;
		int64_t max_j = -1, c;
		int done = 0;
		for (c = 0; !done; ++c) {
			const int in_ring_chunk = (c + 1) * W <= ring;
			int32_t sc[W], incl[W], S[W], M[W], pj[W];
			uint8_t valid[W], Abit[W], Bbit[W];
			int k, kb = -1, all_live = 1, cntA = 0, cntB = 0, a_after_b = 0, seen_b = 0;
			++st->chunks;
			if (!in_ring_chunk) ++st->deep_chunks;
			/* filters (chain.c:252-261) and base score (chain.c:262-273), all lanes at once */
			for (k = 0; k < W; ++k) {
				const int64_t j = i - 1 - c * W - k;
				valid[k] = 0; sc[k] = INT_MIN; pj[k] = -1;
				if (j < u0) { all_live = 0; continue; }
				{
					const uint64_t d64 = ri - a[j].x;
					const int32_t dq = qi - (int32_t)a[j].y;
					const int same = sidi == SEG_OF(a[j].y);
					int64_t dr;
					int32_t dd, s0;
					if (d64 > maxx) { all_live = 0; continue; }
					++st->lane_evals;
					dr = (int64_t)d64;
					pj[k] = p[j];
					if ((same && dr == 0) || dq <= 0) continue;
					if ((same && dq > par->max_dist_y) || dq > par->max_dist_x) continue;
					dd = (int32_t)(dr > dq ? dr - dq : dq - dr);
					if (same && dd > par->bw) continue;
					if (par->n_segs > 1 && !par->is_cdna && same && dr > par->max_dist_y) continue;
					s0 = (int32_t)(dq < dr ? dq : dr);
					if (s0 > span) s0 = span;
					sc[k] = gap_cost_score(s0, dd, dr, dq, same, par->is_cdna, avg) + f[j];
					valid[k] = 1;
				}
			}
			/* derivation 3: write every valid lane's mark first */
			for (k = 0; k < W; ++k) {
				if (valid[k] && pj[k] >= 0) {
					const int64_t tgt = pj[k];
					if (i - tgt <= ring) tl[tgt % ring] = tag;      /* target still in the ring */
					else tg[tgt] = tag;                              /* target only in global memory */
				}
			}
			/* derivation 2: inclusive prefix max, then exclusive seeded with the running max */
			for (k = 0; k < W; ++k) incl[k] = k ? (incl[k - 1] > sc[k] ? incl[k - 1] : sc[k]) : sc[0];
			for (k = 0; k < W; ++k) {
				const int64_t j = i - 1 - c * W - k;
				int32_t excl = k ? (incl[k - 1] > max_f ? incl[k - 1] : max_f) : max_f;
				int marked = 0;
				if (valid[k]) marked = (in_ring_chunk ? tl[j % ring] : tg[j]) == tag;
				Abit[k] = valid[k] && sc[k] > excl;
				Bbit[k] = valid[k] && !Abit[k] && marked;
				if (Bbit[k]) seen_b = 1;
				if (Abit[k] && seen_b) a_after_b = 1;
				cntA += Abit[k]; cntB += Bbit[k];
			}
			/* derivation 4: the n_skip walk */
			if (!a_after_b) {
				/* fast path: every A precedes every B */
				int32_t x = n_skip - cntA; if (x < 0) x = 0;
				int need = par->max_skip - x + 1, seen = 0;
				if (cntB >= need) { for (k = 0; k < W; ++k) if (Bbit[k] && ++seen == need) { kb = k; break; } }
				else n_skip = x + cntB;
			} else {
				int32_t run = n_skip, mn = INT_MAX;
				++st->general_walks;
				for (k = 0; k < W; ++k) {
					run += Bbit[k] - Abit[k];
					S[k] = run;
					mn = run < mn ? run : mn;
					M[k] = mn;
				}
				for (k = 0; k < W; ++k) {
					int32_t x = S[k] - (M[k] < 0 ? M[k] : 0);
					if (Bbit[k] && x > par->max_skip) { kb = k; break; }
				}
				if (kb < 0) n_skip = S[W - 1] - (M[W - 1] < 0 ? M[W - 1] : 0);
			}
			/* the last A lane at or before the break holds the final running max */
			for (k = (kb >= 0 ? kb : W - 1); k >= 0; --k)
				if (Abit[k]) { max_f = sc[k]; max_j = i - 1 - c * W - k; break; }
			if (kb >= 0 || !all_live) done = 1;
		}
		f[i] = max_f; p[i] = (int32_t)max_j;
		v[i] = (max_j >= 0 && v[max_j] > max_f) ? v[max_j] : max_f;
	}
	free(tl);
}

/* Whole batch through the model: K0 (read sums, unit split, singletons) + K1 (units). */
int64_t wm_batch_fpv(const co_params_t *par, int64_t n_reads, const int64_t *off, const co_anchor_t *a,
                     const int32_t *n_segs_per_read, int32_t *f, int32_t *p, int32_t *v, int ring,
                     int64_t *stats_out /* 6 x int64 or NULL */)
{
	const uint64_t maxx = (uint64_t)(int64_t)par->max_dist_x;
	int64_t total = off[n_reads], r;
	int32_t *tg = (int32_t*)calloc(total ? total : 1, 4);
	wm_stats_t st;
	memset(&st, 0, sizeof(st));
	if (ring < W || ring % W) ring = 128;
	for (r = 0; r < n_reads; ++r) {
		int64_t b = off[r], e = off[r + 1], i, u0;
		uint64_t sum = 0;
		float avg;
		co_params_t rp = *par;
		if (n_segs_per_read) rp.n_segs = n_segs_per_read[r];
		if (e <= b) continue;
		for (i = b; i < e; ++i) sum += (uint64_t)SPAN_OF(a[i].y);
		avg = (float)sum / (int64_t)(e - b);
		/* unit offsets are relative to the read in the recurrence; p[] must be read-relative too */
		for (u0 = b, i = b + 1; i <= e; ++i) {
			if (i == e || a[i].x - a[i - 1].x > maxx) {
				++st.units;
				if (i - u0 == 1) {
					++st.singletons;
					f[u0] = v[u0] = SPAN_OF(a[u0].y); p[u0] = -1;
				} else {
					/* run the unit on read-relative indices: shift pointers so index 0 is the read start */
					wm_unit(&rp, avg, a + b, u0 - b, i - b, f + b, p + b, v + b, tg + b, ring, &st);
				}
				u0 = i;
			}
		}
	}
	free(tg);
	if (stats_out) {
		stats_out[0] = st.lane_evals; stats_out[1] = st.chunks; stats_out[2] = st.deep_chunks;
		stats_out[3] = st.general_walks; stats_out[4] = st.units; stats_out[5] = st.singletons;
	}
	return st.lane_evals;
}
