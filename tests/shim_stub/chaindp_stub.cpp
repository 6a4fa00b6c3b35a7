// TEST INFRASTRUCTURE ONLY -- stand-ins for the chaindp_* device calls csrc/fpga_shim.cpp makes, for the sanitizer build of
// the shim (tests/test_shim_sanitizers.py).  No chaining happens here: a "device batch" turns every anchor into one record
// {anchor, p = -4, f = low word of anchor.y} so that the stress driver can check that every read's payload reached its result
// packet intact, in the right slot, exactly once.  Minimizer packets: one "anchor" per minimizer, mini_pos = minimizer.y.
// CHAINDP_ERR_CAPACITY is returned where the real library returns it, so the shim's split-and-retry path runs as well.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/chaindp.h"

struct chaindp_ctx {
	int device; int64_t cap_anchors, cap_reads;
	int64_t n_reads = 0;
	std::vector<int64_t> off;
	std::vector<chaindp_anchor_t> a;
	std::vector<uint64_t> mini_pos;
	std::vector<int64_t> mp_off;
	std::string err;
};
struct chaindp_index { int device; size_t bytes; std::vector<uint8_t> image; };
static std::atomic<int> g_live_indexes{0}, g_index_creates{0};

extern "C" {
int chaindp_device_count(void) { const char *v = getenv("CHAINDP_STUB_GPUS"); return v ? atoi(v) : 2; }
chaindp_ctx_t *chaindp_create(int device, int64_t max_anchors, int64_t max_reads)
{
	chaindp_ctx *c = new chaindp_ctx(); c->device = device; c->cap_anchors = max_anchors; c->cap_reads = max_reads; return c;
}
void chaindp_destroy(chaindp_ctx_t *c) { delete c; }
const char *chaindp_last_error(const chaindp_ctx_t *c) { return c ? c->err.c_str() : "stub"; }
chaindp_index_t *chaindp_index_create(int device, const void *B, size_t nB, const void *H, size_t nH, const void *V, size_t nV, const void *P, size_t nP)
{
	// touch every byte: a truncated or freed host image shows up under ASan
	uint64_t s = 0;
	const void *src[4] = {B, H, V, P}; const size_t n[4] = {nB, nH, nV, nP};
	for (int k = 0; k < 4; ++k) for (size_t i = 0; i < n[k]; ++i) s += ((const uint8_t*)src[k])[i];
	chaindp_index *ix = new chaindp_index(); ix->device = device; ix->bytes = nB + nH + nV + nP + (s & 0);
	ix->image.assign((const uint8_t*)B, (const uint8_t*)B + nB);        // the "device copy": read by every lookup, freed by the destroy
	++g_live_indexes; ++g_index_creates;
	return ix;
}
void chaindp_index_destroy(chaindp_index_t *ix) { if (ix) { --g_live_indexes; delete ix; } }
int chaindp_stub_live_indexes(void) { return g_live_indexes.load(); }
int chaindp_stub_index_creates(void) { return g_index_creates.load(); }

int chaindp_upload_gather_ex(chaindp_ctx_t *c, int64_t n_reads, const int64_t *off, const chaindp_anchor_t *const *ra, const int32_t *, int)
{
	if (n_reads > c->cap_reads || off[n_reads] > c->cap_anchors) { c->err = "capacity"; return CHAINDP_ERR_CAPACITY; }
	c->n_reads = n_reads; c->off.assign(off, off + n_reads + 1); c->a.resize((size_t)off[n_reads]);
	for (int64_t r = 0; r < n_reads; ++r) if (off[r + 1] > off[r]) memcpy(&c->a[(size_t)off[r]], ra[r], (size_t)(off[r + 1] - off[r]) * 16);
	c->mini_pos.clear(); c->mp_off.assign((size_t)n_reads + 1, 0);
	return CHAINDP_OK;
}
int chaindp_collect_seeds_gather(chaindp_ctx_t *c, const chaindp_index_t *ix, int, int, int64_t n_reads, const int64_t *mini_off,
                                 const chaindp_anchor_t *const *rm, const uint32_t *, const int32_t *, const int32_t *, int64_t *off,
                                 int32_t *rep_len, int64_t *mini_pos_off)
{
	if (!ix) { c->err = "no index"; return CHAINDP_ERR_ARG; }
	// the lookup reads the device copy for as long as its "kernels" run: an index destroyed under a batch in flight (the other
	// context of the GPU replacing it) is a use-after-free that ASan reports here
	uint64_t touch = 0;
	for (size_t i = 0; i < ix->image.size(); i += 64) touch += ix->image[i];
	usleep(200);
	for (size_t i = 0; i < ix->image.size(); i += 64) touch += ix->image[i];
	if (touch == 1) c->err = "";
	// "seeds": CHAINDP_STUB_HITS anchors per minimizer (default 1), so that a batch can exceed the context's capacity
	const char *hv = getenv("CHAINDP_STUB_HITS");
	const int64_t hits = hv ? atoi(hv) : 1;
	if (mini_off[n_reads] * hits > c->cap_anchors || n_reads > c->cap_reads) { c->err = "capacity"; return CHAINDP_ERR_CAPACITY; }
	c->n_reads = n_reads; c->off.assign((size_t)n_reads + 1, 0); c->mp_off.assign((size_t)n_reads + 1, 0);
	c->a.clear(); c->mini_pos.clear();
	for (int64_t r = 0; r < n_reads; ++r) {
		const int64_t n = mini_off[r + 1] - mini_off[r];
		for (int64_t k = 0; k < n; ++k) { for (int64_t h = 0; h < hits; ++h) c->a.push_back(rm[r][k]); c->mini_pos.push_back(rm[r][k].y); }
		c->off[(size_t)r + 1] = (int64_t)c->a.size(); c->mp_off[(size_t)r + 1] = (int64_t)c->mini_pos.size();
		rep_len[r] = (int32_t)n;
	}
	memcpy(off, c->off.data(), (size_t)(n_reads + 1) * 8); memcpy(mini_pos_off, c->mp_off.data(), (size_t)(n_reads + 1) * 8);
	return CHAINDP_OK;
}
int chaindp_download_mini_pos(chaindp_ctx_t *c, uint64_t *mp) { if (!c->mini_pos.empty()) memcpy(mp, c->mini_pos.data(), c->mini_pos.size() * 8); return CHAINDP_OK; }
int chaindp_run(chaindp_ctx_t *, const chaindp_params_t *) { return CHAINDP_OK; }
int chaindp_compact_offsets(chaindp_ctx_t *c, const chaindp_params_t *, int64_t *soff) { memcpy(soff, c->off.data(), (size_t)(c->n_reads + 1) * 8); return CHAINDP_OK; }
static void rec(chaindp_seed_t *d, const chaindp_anchor_t &a) { d->seed = a; d->p = -4; d->f = (int32_t)(uint32_t)a.y; }
int chaindp_scatter_seeds(chaindp_ctx_t *c, int64_t n_reads, chaindp_seed_t *const *dst)
{
	for (int64_t r = 0; r < n_reads; ++r) {
		if (!dst[r]) continue;
		const int64_t n = c->off[(size_t)r + 1] - c->off[(size_t)r];
		for (int64_t k = 0; k < n; ++k) rec(&dst[r][k], c->a[(size_t)(c->off[(size_t)r] + k)]);
		const size_t b = (size_t)n * 24, pad = ((b + 63) & ~(size_t)63) - b;
		if (pad) memset((char*)dst[r] + b, 0, pad);
	}
	return CHAINDP_OK;
}
int chaindp_scatter_mini_pos(chaindp_ctx_t *c, int64_t n_reads, uint64_t *const *dst)
{
	for (int64_t r = 0; r < n_reads; ++r) {
		if (!dst[r]) continue;
		const int64_t n = c->mp_off[(size_t)r + 1] - c->mp_off[(size_t)r];
		if (n) memcpy(dst[r], &c->mini_pos[(size_t)c->mp_off[(size_t)r]], (size_t)n * 8);
		const size_t b = (size_t)n * 8, pad = ((b + 63) & ~(size_t)63) - b;
		if (pad) memset((char*)dst[r] + b, 0, pad);
	}
	return CHAINDP_OK;
}
int chaindp_download_seeds(chaindp_ctx_t *c, int64_t first, int64_t n, chaindp_seed_t *dst) { for (int64_t k = 0; k < n; ++k) rec(&dst[k], c->a[(size_t)(first + k)]); return CHAINDP_OK; }
int chaindp_sync(chaindp_ctx_t *) { return CHAINDP_OK; }
}
