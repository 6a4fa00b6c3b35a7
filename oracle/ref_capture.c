/*
 * ref_capture.c -- harness around the UNMODIFIED reference chain.c.
 * TEST INFRASTRUCTURE ONLY; built only in the container that has
 * /root/reference (oracle/Makefile target `ref`), output in oracle/_ref/.
 *
 * mm_chain_dp_fpga (reference chain.c:218-327) frees f/p/t/v before it
 * returns and hands back only new_seed[].  To obtain the raw f[]/p[]/v[] from
 * the reference without editing it, chain.c is compiled with
 * -Dmalloc=cap_malloc -Dfree=cap_free; the allocation ORDER in the prologue
 * (chain.c:228-233: f, p, t, v, fpga_id, fpga_a) identifies each buffer and
 * cap_free() snapshots f, p and v just before they are released
 * (chain.c:318-321).  This file is compiled WITHOUT those defines.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t x, y; } anchor_t;
struct seed_rec { anchor_t seed; int32_t p, f; };

/* the reference entry point (mmpriv.h:69), from chain.c built with the capture defines */
extern struct seed_rec *mm_chain_dp_fpga(int max_dist_x, int max_dist_y, int bw, int max_skip, int min_sc,
                                         int is_cdna, int n_segs, int64_t n, anchor_t *a, uint32_t *new_i);

static __thread struct {
	int armed, n_alloc;
	void *ptr[6];
	int32_t *f, *p, *v;
	size_t bytes;
} cap;

void *cap_malloc(size_t s)
{
	void *q = malloc(s);
	if (cap.armed && cap.n_alloc < 6) cap.ptr[cap.n_alloc++] = q;
	return q;
}

void cap_free(void *q)
{
	if (cap.armed && q) {
		if (q == cap.ptr[0] && cap.f) memcpy(cap.f, q, cap.bytes);
		else if (q == cap.ptr[1] && cap.p) memcpy(cap.p, q, cap.bytes);
		else if (q == cap.ptr[3] && cap.v) memcpy(cap.v, q, cap.bytes);
	}
	free(q);
}

/* Runs the reference on a private copy of a[] (the reference frees its input,
 * chain.c:322).  f,p,v: int32[n] outputs; seeds: room for n records.
 * Returns new_i. */
uint32_t ref_capture_top(int max_dist_x, int max_dist_y, int bw, int max_skip, int min_sc, int is_cdna, int n_segs,
                         int64_t n, const anchor_t *a, int32_t *f, int32_t *p, int32_t *v, struct seed_rec *seeds)
{
	uint32_t new_i = 0;
	size_t bytes = (size_t)(n > 0 ? n : 0) * sizeof(anchor_t);
	anchor_t *copy = (anchor_t*)malloc(bytes ? bytes : 1);
	struct seed_rec *out;
	memcpy(copy, a, bytes);
	memset(&cap, 0, sizeof(cap));
	cap.f = f, cap.p = p, cap.v = v, cap.bytes = (size_t)(n > 0 ? n : 0) * 4;
	cap.armed = 1;
	out = mm_chain_dp_fpga(max_dist_x, max_dist_y, bw, max_skip, min_sc, is_cdna, n_segs, n, copy, &new_i);
	cap.armed = 0;
	if (seeds && new_i) memcpy(seeds, out, (size_t)new_i * sizeof(struct seed_rec));
	free(out);
	return new_i;
}
