// chaindp_abi.cpp -- host side of the C ABI declared in include/chaindp.h.
// Owns the per-GPU context (stream, HBM buffers, scratch), stages batches and launches the kernels
// of chaindp_kernels.hip / chaindp_compact.hip.  No CPU implementation of the DP exists in this
// library: without a GPU every entry point fails with CHAINDP_ERR_NODEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/chaindp.h"
#include "chaindp_kernels.h"

using chaindp::Params;
using chaindp::Unit;

static thread_local std::string g_create_error;

struct EventSet { hipEvent_t e[3]; int n; int slot0; };  // e[0..n): consecutive kernel boundaries; slot0 = first ms[] index

struct chaindp_ctx {
	int device = -1;
	hipStream_t stream = nullptr;
	int64_t cap_anchors = 0, cap_reads = 0;
	int ring = 128;
	// resident batch
	int64_t n_reads = 0, total = 0, n_seeds = 0;
	bool has_n_segs = false, ran = false, compact_ready = false;
	bool singles_pending = false;    // the last run left f, p, v, flags[] of its singletons to k_fill_singles (chaindp_download runs it)
	chaindp_params_t ran_par{};      // the parameters of that run
	int64_t *d_off = nullptr;
	void *d_a = nullptr;
	int32_t *d_n_segs = nullptr;
	int32_t *d_f = nullptr, *d_p = nullptr, *d_v = nullptr;
	// scratch
	unsigned long long *d_tg = nullptr;   // deep-path marks, tagged with the run epoch (never re-initialised)
	uint32_t epoch = 0;
	unsigned long long *d_sumq = nullptr;
	Unit *d_units = nullptr;
	chaindp::UnitAux *d_unit_aux = nullptr;   // per unit, beside d_units: what k_chain_twin needs to pick it up without further loads
	Unit *d_left = nullptr;               // units the two-per-wave kernel hands over to k_chain_units
	unsigned long long *d_left_cnt = nullptr;   // [0] handed-over count | the twin / quad kernel's queue << 32; [1] count of d_deep; [2] k_chain_dense1's two queues;
	                                            // [3] route: 1 = k_chain_quad took the batch
	Unit *d_deep = nullptr;               // units k_chain_units hands over to its k_chain_dense (scans that keep reaching past the ring)
	int deep_route = 0;                   // test hook: 1 k_chain_dense, 2 k_chain_dense1 whatever the batch looks like
	int deep_eager = 0;                   // test hook: hand over any unit with a few deep scans, whatever its length
	bool deep_handover = true;            // CHAINDP_NO_DEEP_HANDOVER (diagnostic / A-B): every unit stays in the launch that took it
	bool use_quad = false;                // CHAINDP_QUAD=1 / chaindp_debug_set_quad (A/B, tests): one-table batches of ordinary units four per wave
	                                      // (k_chain_quad: correct, measured slower than k_chain_twin -- DESIGN.md section 6 -- so off by default)
	int twin_force_left = 0;              // CHAINDP_TWIN_FORCE_LEFT / chaindp_debug_set_twin_handover (tests): 1 k_chain_twin hands every unit
	                                      // over untouched, 2 after its first tile (k_chain_units resumes there); the variable is read once, at chaindp_create
	int variant = 0;                      // 0: k_chain_twin + k_chain_units for the rest; 1: k_chain_units, general variant; 2: k_chain_units only
	unsigned long long *d_counters = nullptr;
	chaindp::PrepassScratch pre = {nullptr, nullptr, nullptr, nullptr, nullptr};
	chaindp::CompactScratch cmp = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	chaindp::BottomScratch bot = {};
	bool bot_ready = false, seed_ready = false;   // first-use allocations complete
	std::vector<void*> bot_allocs;
	uint16_t *d_lut = nullptr;
	void **d_ptrs = nullptr;         // per-read host pointers for the gather / scatter kernels
	size_t ptr_cap = 0;
	size_t lut_bytes = 0;
	bool use_lut = true;
	// compaction (allocated on first use)
	int32_t *d_first_child = nullptr;
	unsigned int *d_twin_queue = nullptr;   // k_chain_twin's eight grab counters, a cache line apart (2 KB)
	int64_t *d_seeds_off = nullptr;
	void *d_seeds = nullptr;
	// seed collection (allocated on first use, grown with the batch)
	chaindp::SeedScratch seed = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	void *d_mini = nullptr;
	int64_t *d_mini_off = nullptr, *d_mp_off = nullptr;
	uint32_t *d_bid = nullptr;
	int32_t *d_qlen = nullptr, *d_rep_len = nullptr;
	unsigned long long *d_mini_pos = nullptr;
	int64_t seed_cap_mini = 0, n_mini_pos = 0;
	int seed_max_n = -1, seed_max_n2 = -1; // largest reads the two configurations of the LDS sort take on this device
	int seed_lab_cap = 0;                  // digits k_seed_sort_huge keeps in LDS
	// chains to hits (allocated on first use, grown with the batch; freed with bot_allocs)
	void *d_regs = nullptr, *d_reg_counts = nullptr, *d_ref_len = nullptr, *d_mp_up = nullptr;
	size_t regs_cap = 0, reg_counts_cap = 0, ref_len_cap = 0, mp_up_cap = 0;
	uint32_t *d_rhash = nullptr;
	int32_t *d_rqlen = nullptr;
	int64_t *d_regs_off = nullptr, *d_mp_off_up = nullptr;
	unsigned long long *d_sum_k = nullptr;
	int64_t bot_n_reads = -1, bot_n_chains = 0, bot_n_b = 0;   // what the last chaindp_backtrack left resident (-1: nothing of this batch)
	bool mp_resident = false;                                  // this batch's mini_pos are on the device (it came from chaindp_collect_seeds)
	// profiling
	bool prof = false;
	std::vector<EventSet> pending;
	double ms[4] = {0, 0, 0, 0};
	int64_t launches[4] = {0, 0, 0, 0};
	int64_t stats[4] = {0, 0, 0, 0};
	std::string err;
};

#define HIP_TRY(ctx, call)                                                                         \
	do {                                                                                           \
		hipError_t e_ = (call);                                                                    \
		if (e_ != hipSuccess) {                                                                    \
			(ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
			return CHAINDP_ERR_HIP;                                                                \
		}                                                                                          \
	} while (0)

static Params to_params(const chaindp_params_t *p)
{
	Params q;
	q.max_dist_x = p->max_dist_x; q.max_dist_y = p->max_dist_y; q.bw = p->bw; q.max_skip = p->max_skip;
	q.min_sc = p->min_sc; q.is_cdna = p->is_cdna; q.n_segs = p->n_segs;
	return q;
}

static int check_params(chaindp_ctx *ctx, const chaindp_params_t *par)
{
	if (!par) { ctx->err = "params is NULL"; return CHAINDP_ERR_ARG; }
	// the reference compares unsigned differences against these after an int -> u64 conversion
	// (chain.c:252); negative values would silently mean "unbounded", refuse them instead
	if (par->max_dist_x < 0 || par->max_dist_y < 0 || par->bw < 0) {
		ctx->err = "max_dist_x, max_dist_y and bw must be >= 0";
		return CHAINDP_ERR_ARG;
	}
	return CHAINDP_OK;
}

extern "C" int chaindp_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" const char *chaindp_last_error(const chaindp_ctx_t *ctx)
{
	return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" void chaindp_destroy(chaindp_ctx_t *ctx)
{
	if (!ctx) return;
	if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	for (auto &es : ctx->pending) for (int k = 0; k < es.n; ++k) (void)hipEventDestroy(es.e[k]);
	void *bufs[] = {ctx->d_twin_queue, ctx->d_off, ctx->d_a, ctx->d_n_segs, ctx->d_f, ctx->d_p, ctx->d_v, ctx->d_tg, ctx->d_sumq, ctx->d_units,
	                ctx->d_counters, ctx->d_unit_aux, ctx->d_left, ctx->d_left_cnt, ctx->d_deep, ctx->pre.start_mask, ctx->pre.single_mask, ctx->pre.emit_mask, ctx->pre.block_cnt, ctx->pre.tile_tmp, ctx->pre.units_tmp, ctx->pre.hist, ctx->pre.block_reads, ctx->d_lut, ctx->d_ptrs, ctx->cmp.flags, ctx->cmp.block_cnt, ctx->cmp.tile_tmp, ctx->cmp.n_seeds, ctx->cmp.sub, ctx->d_first_child, ctx->d_seeds_off, ctx->d_seeds};
	for (void *b : bufs) if (b) (void)hipFree(b);
	for (void *b : ctx->bot_allocs) if (b) (void)hipFree(b);
	void *sbufs[] = {ctx->seed.kept, ctx->seed.used, ctx->seed.src, ctx->seed.mstate, ctx->seed.tile_tmp, ctx->seed.totals, ctx->seed.stacks,
	                 ctx->d_mini, ctx->d_mini_off, ctx->d_mp_off, ctx->d_bid, ctx->d_qlen, ctx->d_rep_len, ctx->d_mini_pos};
	for (void *b : sbufs) if (b) (void)hipFree(b);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

extern "C" chaindp_ctx_t *chaindp_create(int device, int64_t max_anchors, int64_t max_reads)
{
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { g_create_error = "no HIP device visible"; return nullptr; }
	if (device < 0 || device >= n_dev || max_anchors < 0 || max_reads < 0 || max_anchors > 0x7fffffff || max_reads > 0x7fffffff) {
		g_create_error = "bad device index or capacity (at most 2^31-1 anchors and reads per batch)";
		return nullptr;
	}
	chaindp_ctx *ctx = new chaindp_ctx();
	ctx->device = device;
	ctx->cap_anchors = max_anchors > 0 ? max_anchors : 1;
	ctx->cap_reads = max_reads > 0 ? max_reads : 1;
	const size_t na = (size_t)ctx->cap_anchors, nr = (size_t)ctx->cap_reads;
	hipError_t e = hipSetDevice(device);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_off, (nr + 1) * 8);
	if (e == hipSuccess) e = hipMalloc(&ctx->d_a, na * 16);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_n_segs, nr * 4);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_f, na * 4);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_twin_queue, 8 * 64 * sizeof(unsigned int));
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_p, na * 4);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_v, na * 4);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_tg, na * 8);
	if (e == hipSuccess) e = hipMemset(ctx->d_tg, 0, na * 8);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_sumq, nr * 8);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_units, (na / 2 + 1) * sizeof(Unit));
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_unit_aux, (na / 2 + 1) * sizeof(chaindp::UnitAux));
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_counters, 2 * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_left, (na / 2 + 1) * sizeof(Unit));
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_left_cnt, 4 * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_deep, (na / 64 + 2) * sizeof(Unit));    // a unit is handed over after its first 64-anchor tile at the earliest (test mode), with anchors to go
	size_t flags_bytes0 = 0, cblocks_bytes0 = 0;
	chaindp::compact_scratch_bytes(ctx->cap_anchors, &flags_bytes0, &cblocks_bytes0);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_first_child, na * 4);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->cmp.flags, flags_bytes0);
	size_t mask_bytes = 0, blocks_bytes = 0;
	chaindp::prepass_scratch_bytes(ctx->cap_anchors, &mask_bytes, &blocks_bytes);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->pre.start_mask, mask_bytes);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->pre.single_mask, mask_bytes);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->pre.emit_mask, mask_bytes);
	ctx->cmp.single_mask = ctx->pre.single_mask; ctx->cmp.emit_mask = ctx->pre.emit_mask;
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->pre.block_cnt, blocks_bytes);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->pre.tile_tmp, blocks_bytes);
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->pre.units_tmp, (na / 2 + 1) * sizeof(Unit));
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->pre.hist, (2 * 128 + 2) * sizeof(unsigned int));
	ctx->pre.key_range = ctx->pre.hist ? ctx->pre.hist + 2 * 128 : nullptr;
	if (e == hipSuccess) e = hipMalloc((void**)&ctx->pre.block_reads, blocks_bytes);   // 8 B per block, like the counters
	ctx->cmp.block_reads = ctx->pre.block_reads;
	ctx->use_quad = getenv("CHAINDP_QUAD") != nullptr;
	ctx->deep_handover = getenv("CHAINDP_NO_DEEP_HANDOVER") == nullptr;      // diagnostic switches are read here, once per context:
	if (const char *v = getenv("CHAINDP_TWIN_FORCE_LEFT")) ctx->twin_force_left = atoi(v) == 2 ? 2 : 1;   // never on the launch path (contexts run from several host threads)
	if (e != hipSuccess) {
		g_create_error = std::string("chaindp_create: ") + hipGetErrorString(e);
		chaindp_destroy(ctx);
		return nullptr;
	}
	return ctx;
}

extern "C" int chaindp_set_ring(chaindp_ctx_t *ctx, int ring)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (ring != 128 && ring != 256 && ring != 512) { ctx->err = "ring must be 128, 256 or 512"; return CHAINDP_ERR_ARG; }
	ctx->ring = ring;
	return CHAINDP_OK;
}

extern "C" int chaindp_set_variant(chaindp_ctx_t *ctx, int force_general)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (force_general < 0 || force_general > 2) { ctx->err = "variant must be 0 (two units per wave + the rest), 1 (general) or 2 (one unit per wave)"; return CHAINDP_ERR_ARG; }
	ctx->use_lut = force_general != 1;
	ctx->variant = force_general;
	return CHAINDP_OK;
}

extern "C" int chaindp_set_profiling(chaindp_ctx_t *ctx, int on)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	ctx->prof = on != 0;
	return CHAINDP_OK;
}

// Launch prepass + chain DP on `st` for a batch described by device pointers, using ctx's scratch.
static int run_on_stream(chaindp_ctx *ctx, const chaindp_params_t *par, int64_t n_reads, int64_t total,
                         const int64_t *d_off, const void *d_a, const int32_t *d_n_segs,
                         int32_t *d_f, int32_t *d_p, int32_t *d_v, hipStream_t st)
{
	int rc = check_params(ctx, par);
	if (rc) return rc;
	if (n_reads < 0 || total < 0) { ctx->err = "negative batch size"; return CHAINDP_ERR_ARG; }
	if (n_reads > ctx->cap_reads || total > ctx->cap_anchors) {
		ctx->err = "batch exceeds the capacity the context was created with";
		return CHAINDP_ERR_CAPACITY;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const Params q = to_params(par);
	EventSet es; es.n = 0; es.slot0 = 0;
	if (ctx->prof) {
		for (int k = 0; k < 3; ++k) HIP_TRY(ctx, hipEventCreate(&es.e[k]));
		es.n = 3;
	}
	if (ctx->prof) HIP_TRY(ctx, hipEventRecord(es.e[0], st));
	HIP_TRY(ctx, chaindp::launch_prepass(st, q, n_reads, total, d_off, d_a, ctx->d_sumq, ctx->d_units, ctx->d_counters, ctx->pre, ctx->d_unit_aux, d_n_segs, ctx->d_left_cnt));
	if (ctx->prof) HIP_TRY(ctx, hipEventRecord(es.e[1], st));
	// per-read gap-cost table for the fast variant (skipped when the table would not apply)
	uint16_t *lut = nullptr;
	int lut_stride = 0;
	if (ctx->use_lut && !q.is_cdna && q.bw <= CHAINDP_LUT_MAX_BW && n_reads > 0) {
		lut_stride = (q.bw + 1 + 7) & ~7;
		const size_t need = (size_t)n_reads * lut_stride * sizeof(uint16_t);
		if (need > ctx->lut_bytes) {
			HIP_TRY(ctx, hipStreamSynchronize(st));
			if (ctx->d_lut) HIP_TRY(ctx, hipFree(ctx->d_lut));
			ctx->d_lut = nullptr; ctx->lut_bytes = 0;
			HIP_TRY(ctx, hipMalloc((void**)&ctx->d_lut, need));
			ctx->lut_bytes = need;
		}
		lut = ctx->d_lut;
		HIP_TRY(ctx, chaindp::launch_lut(st, q, n_reads, d_off, ctx->d_sumq, lut_stride, lut));
	}
	if (++ctx->epoch == 0) {                                       // 2^32 runs: start the mark epochs over
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_tg, 0, (size_t)ctx->cap_anchors * 8, st));
		ctx->epoch = 1;
	}
	// (d_left_cnt -- hand-over counts and the twin kernel's queue word -- was zeroed by the prepass' first kernel)
	Unit *const deep = ctx->deep_handover ? ctx->d_deep : nullptr;
	unsigned int *const deep_cnt = (unsigned int*)(ctx->d_left_cnt + 1);
	if (ctx->variant == 0 && lut) {
		// ordinary units two per wave; what that kernel hands over (and nothing else) goes through k_chain_units
		// (first_child[] is initialised by the DP kernels themselves, per tile: no batch-wide memset)
		// four units per wave where the whole batch has one cost table, else two per wave: both are launched, the device decides
		unsigned int *const route = (unsigned int*)(ctx->d_left_cnt + 3);
		Params qt = q;                          // with per-read segment counts the units' UnitAux flags say which reads are multi-segment
		if (d_n_segs) qt.n_segs = 1;            // (the batch-wide count is not used then, as in k_chain_units)
		if (ctx->use_quad)
			HIP_TRY(ctx, chaindp::launch_chain_quad(st, qt, total / 2, d_a, lut, lut_stride, ctx->d_units, ctx->d_unit_aux, ctx->d_counters, ctx->pre.key_range,
			                                        d_f, d_p, d_v, ctx->d_first_child, ctx->cmp.flags, ctx->d_left, (unsigned int*)ctx->d_left_cnt,
			                                        (unsigned int*)ctx->d_left_cnt + 1, route, ctx->twin_force_left, total));
		HIP_TRY(ctx, chaindp::launch_chain_twin(st, qt, total / 2, d_off, d_a, ctx->d_sumq, lut, lut_stride, ctx->d_units, ctx->d_counters,
		                                        d_f, d_p, d_v, ctx->d_first_child, ctx->cmp.flags, ctx->d_left, (unsigned int*)ctx->d_left_cnt,
		                                        ctx->twin_force_left, total, ctx->d_unit_aux, route, ctx->d_twin_queue));
		const int64_t left_grid = total / 2 < 32768 ? total / 2 : 32768;
		HIP_TRY(ctx, chaindp::launch_chain(st, ctx->ring, q, left_grid, d_off, d_a, d_n_segs, ctx->d_sumq, lut, lut_stride, ctx->d_left,
		                                   ctx->d_left_cnt, d_f, d_p, d_v, ctx->d_tg, ctx->epoch, ctx->d_first_child, ctx->cmp.flags,
		                                   ctx->d_units, ctx->d_counters, deep, deep_cnt, ctx->pre.hist + CHAINDP_LONG_UNIT_CLASS, ctx->deep_eager, ctx->deep_route));
	} else
		HIP_TRY(ctx, chaindp::launch_chain(st, ctx->ring, q, total / 2, d_off, d_a, d_n_segs, ctx->d_sumq, lut, lut_stride, ctx->d_units,
		                                   ctx->d_counters, d_f, d_p, d_v, ctx->d_tg, ctx->epoch, ctx->d_first_child, ctx->cmp.flags,
		                                   nullptr, nullptr, deep, deep_cnt, ctx->pre.hist + CHAINDP_LONG_UNIT_CLASS, ctx->deep_eager, ctx->deep_route));
	// units whose scans kept reaching past the ring (dense repeats): redone by k_chain_dense
	if (deep && lut) {
		HIP_TRY(ctx, chaindp::launch_chain_dense(st, q, total / 64 + 1, d_off, d_a, lut, lut_stride, ctx->d_deep, ctx->d_left_cnt + 1,
		                                         d_f, d_p, d_v, ctx->d_first_child, ctx->cmp.flags, ctx->pre.hist + CHAINDP_LONG_UNIT_CLASS, ctx->deep_route,
		                                           (unsigned int*)(ctx->d_left_cnt + 3) + 1));   // (the word behind the route flag: zeroed with it)
		HIP_TRY(ctx, chaindp::launch_chain_dense16(st, q, total / 64 + 1, d_off, d_a, lut, lut_stride, ctx->d_deep, ctx->d_left_cnt + 1,
		                                           d_f, d_p, d_v, ctx->d_first_child, ctx->cmp.flags, ctx->pre.hist + CHAINDP_LONG_UNIT_CLASS, ctx->deep_route,
		                                           (unsigned int*)(ctx->d_left_cnt + 3) + 1));   // (the word behind the route flag: zeroed with it)
		HIP_TRY(ctx, chaindp::launch_chain_dense1(st, q, total / 64 + 1, d_off, d_a, lut, lut_stride, ctx->d_deep, ctx->d_left_cnt + 1,
		                                          ctx->pre.hist + CHAINDP_LONG_UNIT_CLASS, ctx->deep_route, (unsigned int*)(ctx->d_left_cnt + 2), d_f, d_p, d_v, ctx->d_first_child, ctx->cmp.flags));
	}
	if (ctx->prof) { HIP_TRY(ctx, hipEventRecord(es.e[2], st)); ctx->pending.push_back(es); }
	ctx->stats[2] = total; ctx->stats[3] = n_reads;
	return CHAINDP_OK;
}

extern "C" int chaindp_upload(chaindp_ctx_t *ctx, int64_t n_reads, const int64_t *off, const chaindp_anchor_t *a,
                              const int32_t *n_segs_per_read)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (n_reads < 0 || !off || (n_reads > 0 && off[0] != 0)) { ctx->err = "bad offsets"; return CHAINDP_ERR_ARG; }
	const int64_t total = n_reads > 0 ? off[n_reads] : 0;
	if (total < 0 || (total > 0 && !a)) { ctx->err = "bad anchors"; return CHAINDP_ERR_ARG; }
	if (n_reads > ctx->cap_reads || total > ctx->cap_anchors) {
		ctx->err = "batch exceeds the capacity the context was created with";
		return CHAINDP_ERR_CAPACITY;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_off, off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
	if (total) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_a, a, (size_t)total * 16, hipMemcpyHostToDevice, ctx->stream));
	ctx->has_n_segs = n_segs_per_read != nullptr;
	if (n_segs_per_read && n_reads)
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_n_segs, n_segs_per_read, (size_t)n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ctx->n_reads = n_reads; ctx->total = total; ctx->ran = false; ctx->bot_n_reads = -1; ctx->mp_resident = false;
	return CHAINDP_OK;
}

extern "C" int chaindp_run(chaindp_ctx_t *ctx, const chaindp_params_t *par)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	int rc = run_on_stream(ctx, par, ctx->n_reads, ctx->total, ctx->d_off, ctx->d_a, ctx->has_n_segs ? ctx->d_n_segs : nullptr,
	                       ctx->d_f, ctx->d_p, ctx->d_v, ctx->stream);
	if (rc == CHAINDP_OK) { ctx->ran = true; ctx->singles_pending = true; ctx->ran_par = *par; }
	return rc;
}

extern "C" int chaindp_run_device(chaindp_ctx_t *ctx, const chaindp_params_t *par, int64_t n_reads, int64_t total_anchors,
                                  const void *d_off, const void *d_a, const void *d_n_segs,
                                  void *d_f, void *d_p, void *d_v, void *stream)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (!d_off || (total_anchors > 0 && (!d_a || !d_f || !d_p || !d_v))) { ctx->err = "NULL device pointer"; return CHAINDP_ERR_ARG; }
	hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
	const int rc = run_on_stream(ctx, par, n_reads, total_anchors, (const int64_t*)d_off, d_a, (const int32_t*)d_n_segs,
	                             (int32_t*)d_f, (int32_t*)d_p, (int32_t*)d_v, st);
	if (rc != CHAINDP_OK) return rc;
	// the caller reads its own arrays: the singletons' entries are written right away
	HIP_TRY(ctx, chaindp::launch_fill_singles(st, to_params(par), total_anchors, d_a, ctx->pre, (int32_t*)d_f, (int32_t*)d_p, (int32_t*)d_v, ctx->cmp.flags));
	return CHAINDP_OK;
}

extern "C" int chaindp_sync(chaindp_ctx_t *ctx)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CHAINDP_OK;
}

extern "C" int chaindp_download(chaindp_ctx_t *ctx, int32_t *f, int32_t *p, int32_t *v)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (!ctx->ran) { ctx->err = "chaindp_download before chaindp_run"; return CHAINDP_ERR_ARG; }
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const size_t bytes = (size_t)ctx->total * 4;
	if (ctx->singles_pending) {          // (the compaction works from the prepass' masks: only this call looks at the singletons' f, p, v)
		HIP_TRY(ctx, chaindp::launch_fill_singles(ctx->stream, to_params(&ctx->ran_par), ctx->total, ctx->d_a, ctx->pre, ctx->d_f, ctx->d_p, ctx->d_v, ctx->cmp.flags));
		ctx->singles_pending = false;
	}
	if (bytes) {
		if (f) HIP_TRY(ctx, hipMemcpyAsync(f, ctx->d_f, bytes, hipMemcpyDeviceToHost, ctx->stream));
		if (p) HIP_TRY(ctx, hipMemcpyAsync(p, ctx->d_p, bytes, hipMemcpyDeviceToHost, ctx->stream));
		if (v) HIP_TRY(ctx, hipMemcpyAsync(v, ctx->d_v, bytes, hipMemcpyDeviceToHost, ctx->stream));
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CHAINDP_OK;
}

extern "C" int chaindp_chain_batch(chaindp_ctx_t *ctx, const chaindp_params_t *par, int64_t n_reads, const int64_t *off,
                                   const chaindp_anchor_t *a, const int32_t *n_segs_per_read,
                                   int32_t *f, int32_t *p, int32_t *v)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	int rc = check_params(ctx, par);
	if (rc) return rc;
	if ((rc = chaindp_upload(ctx, n_reads, off, a, n_segs_per_read)) != CHAINDP_OK) return rc;
	if ((rc = chaindp_run(ctx, par)) != CHAINDP_OK) return rc;
	return chaindp_download(ctx, f, p, v);
}

// launches the compaction kernels on the context's stream (asynchronous)
static int compact_launch(chaindp_ctx *ctx, const chaindp_params_t *par)
{
	int rc = check_params(ctx, par);
	if (rc) return rc;
	if (!ctx->ran) { ctx->err = "compaction before chaindp_run"; return CHAINDP_ERR_ARG; }
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (!ctx->compact_ready) {
		// first use: every buffer is allocated into a local and committed only when all of them exist, so that an
		// out-of-memory here leaves the context as it was (the next call tries again) instead of half-initialised
		const size_t na = (size_t)ctx->cap_anchors, nr = (size_t)ctx->cap_reads;
		size_t flags_bytes = 0, blocks_bytes = 0;
		chaindp::compact_scratch_bytes(ctx->cap_anchors, &flags_bytes, &blocks_bytes);
		void *nb[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
		const size_t sz[6] = {(nr + 1) * 8, ctx->d_seeds ? 0 : na * sizeof(chaindp_seed_t) + 16, blocks_bytes, blocks_bytes, 8, blocks_bytes * 32};
		hipError_t e = hipSuccess;
		for (int k = 0; k < 6 && e == hipSuccess; ++k) if (sz[k]) e = hipMalloc(&nb[k], sz[k]);
		if (e != hipSuccess) {
			for (void *b : nb) if (b) (void)hipFree(b);
			ctx->err = std::string("compaction buffers: ") + hipGetErrorString(e);
			return CHAINDP_ERR_HIP;
		}
		ctx->d_seeds_off = (int64_t*)nb[0];
		if (nb[1]) ctx->d_seeds = nb[1];                        // (seed collection may have made it already)
		ctx->cmp.block_cnt = (unsigned long long*)nb[2]; ctx->cmp.tile_tmp = (unsigned long long*)nb[3];
		ctx->cmp.n_seeds = (unsigned long long*)nb[4]; ctx->cmp.sub = (uint32_t*)nb[5];
		ctx->compact_ready = true;
	}
	EventSet es; es.n = 0; es.slot0 = 2;
	if (ctx->prof) {
		for (int k = 0; k < 2; ++k) HIP_TRY(ctx, hipEventCreate(&es.e[k]));
		es.n = 2;
		HIP_TRY(ctx, hipEventRecord(es.e[0], ctx->stream));
	}
	HIP_TRY(ctx, chaindp::launch_compact(ctx->stream, to_params(par), ctx->n_reads, ctx->total, ctx->d_off, ctx->d_a, ctx->d_f, ctx->d_p,
	                                     ctx->d_v, ctx->d_first_child, ctx->d_seeds_off, ctx->d_seeds, ctx->cmp));
	if (ctx->prof) { HIP_TRY(ctx, hipEventRecord(es.e[1], ctx->stream)); ctx->pending.push_back(es); }
	return CHAINDP_OK;
}

static int compact_on_device(chaindp_ctx *ctx, const chaindp_params_t *par, int64_t *seeds_off)
{
	if (!seeds_off) { ctx->err = "NULL output"; return CHAINDP_ERR_ARG; }
	int rc = compact_launch(ctx, par);
	if (rc) return rc;
	HIP_TRY(ctx, hipMemcpyAsync(seeds_off, ctx->d_seeds_off, (size_t)(ctx->n_reads + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ctx->n_seeds = seeds_off[ctx->n_reads];
	return CHAINDP_OK;
}

template <typename T>
static hipError_t bot_alloc(chaindp_ctx *ctx, T *&p, size_t bytes)
{
	void *q = nullptr;
	hipError_t e = hipMalloc(&q, bytes ? bytes : 8);
	if (e == hipSuccess) { ctx->bot_allocs.push_back(q); p = (T*)q; }
	return e;
}

extern "C" int chaindp_backtrack(chaindp_ctx_t *ctx, const chaindp_params_t *par, int min_cnt,
                                 int64_t *chains_off, uint64_t *u, int64_t *b_off, chaindp_anchor_t *b)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	int rc = check_params(ctx, par);
	if (rc) return rc;
	if (!ctx->ran || !ctx->d_seeds) { ctx->err = "chaindp_backtrack needs a completed run and compaction"; return CHAINDP_ERR_ARG; }
	if (!chains_off || !b_off) { ctx->err = "NULL output"; return CHAINDP_ERR_ARG; }
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	// the record count of the last compaction (it may have been launched asynchronously by chaindp_run_full)
	unsigned long long n_seeds = 0;
	HIP_TRY(ctx, hipMemcpyAsync(&n_seeds, ctx->cmp.n_seeds, 8, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	const int64_t m = ctx->total > 0 && ctx->n_reads > 0 ? (int64_t)(uint32_t)n_seeds : 0;
	ctx->n_seeds = m;
	if (!ctx->bot_ready) {
		// first use.  An out-of-memory half way leaves the context as it was (what this attempt allocated is freed again and the
		// next call tries anew) instead of half-initialised with kernels launched on null scratch pointers.
		const size_t M = (size_t)ctx->cap_anchors, R = (size_t)ctx->cap_reads, NB = M / 1024 + 2;
		chaindp::BottomScratch &s = ctx->bot;
		const size_t first = ctx->bot_allocs.size();
		hipError_t e = hipSuccess;
#define BOT_ALLOC(field, bytes) if (e == hipSuccess) e = bot_alloc(ctx, s.field, (bytes))
		BOT_ALLOC(has, M);
		BOT_ALLOC(owner, M * 4); BOT_ALLOC(end_rec, M * 4); BOT_ALLOC(ccnt, M * 4); BOT_ALLOC(kpos, M * 4); BOT_ALLOC(bpos, M * 4);
		BOT_ALLOC(c_src, M * 4); BOT_ALLOC(c_dst, M * 4);
		BOT_ALLOC(key, M * 8); BOT_ALLOC(skey, M * 8); BOT_ALLOC(cu, M * 8); BOT_ALLOC(u_tmp, M * 8); BOT_ALLOC(u_out, M * 8);
		BOT_ALLOC(b_tmp, M * 16); BOT_ALLOC(b_out, M * 16); BOT_ALLOC(w, M * 16);
		BOT_ALLOC(stacks, (M / 64 + 2 * R + 4) * 12);
		BOT_ALLOC(block_cnt, (NB > R + 2 ? NB : R + 2) * 8); BOT_ALLOC(tile_tmp, (NB > R + 2 ? NB : R + 2) * 8);
		BOT_ALLOC(read_tot, (R + 2) * 8); BOT_ALLOC(total, 8);
		BOT_ALLOC(ends_off, (R + 2) * 8); BOT_ALLOC(chains_off, (R + 2) * 8); BOT_ALLOC(b_off, (R + 2) * 8);
#undef BOT_ALLOC
		if (e != hipSuccess) {
			for (size_t k = first; k < ctx->bot_allocs.size(); ++k) if (ctx->bot_allocs[k]) (void)hipFree(ctx->bot_allocs[k]);
			ctx->bot_allocs.resize(first);
			ctx->bot = chaindp::BottomScratch{};
			ctx->err = std::string("backtrack buffers: ") + hipGetErrorString(e);
			return CHAINDP_ERR_HIP;
		}
		ctx->bot_ready = true;
	}
	EventSet es; es.n = 0; es.slot0 = 3;
	if (ctx->prof) {
		for (int k = 0; k < 2; ++k) HIP_TRY(ctx, hipEventCreate(&es.e[k]));
		es.n = 2;
		HIP_TRY(ctx, hipEventRecord(es.e[0], ctx->stream));
	}
	HIP_TRY(ctx, chaindp::launch_backtrack(ctx->stream, min_cnt, par->min_sc, ctx->n_reads, ctx->cap_anchors, ctx->d_seeds_off, ctx->d_seeds,
	                                       ctx->cmp.n_seeds, ctx->bot, m));
	if (ctx->prof) { HIP_TRY(ctx, hipEventRecord(es.e[1], ctx->stream)); ctx->pending.push_back(es); }
	const size_t ob = (size_t)(ctx->n_reads > 0 ? ctx->n_reads + 1 : 1) * 8;
	HIP_TRY(ctx, hipMemcpyAsync(chains_off, ctx->bot.chains_off, ob, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(b_off, ctx->bot.b_off, ob, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	const int64_t n_c = ctx->n_reads > 0 ? chains_off[ctx->n_reads] : 0, n_b = ctx->n_reads > 0 ? b_off[ctx->n_reads] : 0;
	ctx->bot_n_reads = ctx->n_reads; ctx->bot_n_chains = n_c; ctx->bot_n_b = n_b;
	if (u && n_c > 0) HIP_TRY(ctx, hipMemcpyAsync(u, ctx->bot.u_out, (size_t)n_c * 8, hipMemcpyDeviceToHost, ctx->stream));
	if (b && n_b > 0) HIP_TRY(ctx, hipMemcpyAsync(b, ctx->bot.b_out, (size_t)n_b * 16, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CHAINDP_OK;
}

// grow-only device buffer owned by the context (freed with the backtrack allocations)
static hipError_t regs_grow(chaindp_ctx *ctx, void *&p, size_t &cap, size_t need)
{
	if (need <= cap && p) return hipSuccess;
	void *q = nullptr;
	hipError_t e = hipMalloc(&q, need ? need + need / 4 : 8);
	if (e != hipSuccess) return e;
	for (void *&old : ctx->bot_allocs) if (old == p && p) { (void)hipFree(p); old = nullptr; }
	ctx->bot_allocs.push_back(q);
	p = q; cap = need + need / 4;
	return hipSuccess;
}

static int regs_per_read_buffers(chaindp_ctx *ctx)
{
	if (ctx->d_rhash) return CHAINDP_OK;
	const size_t R = (size_t)ctx->cap_reads + 2;
	HIP_TRY(ctx, bot_alloc(ctx, ctx->d_rhash, R * 4)); HIP_TRY(ctx, bot_alloc(ctx, ctx->d_rqlen, R * 4));
	HIP_TRY(ctx, bot_alloc(ctx, ctx->d_regs_off, R * 8)); HIP_TRY(ctx, bot_alloc(ctx, ctx->d_mp_off_up, R * 8));
	HIP_TRY(ctx, bot_alloc(ctx, ctx->d_sum_k, R * 8));
	return CHAINDP_OK;
}

extern "C" int chaindp_gen_regs(chaindp_ctx_t *ctx, const uint32_t *hash, const int32_t *qlen, chaindp_reg_t *regs)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (ctx->bot_n_reads < 0 || ctx->bot_n_reads != ctx->n_reads || !ctx->bot.has) { ctx->err = "chaindp_gen_regs needs the chains of a chaindp_backtrack on this batch"; return CHAINDP_ERR_ARG; }
	const int64_t R = ctx->bot_n_reads, n_c = ctx->bot_n_chains;
	if (R > 0 && (!hash || !qlen)) { ctx->err = "NULL hash or qlen"; return CHAINDP_ERR_ARG; }
	if (n_c > 0 && !regs) { ctx->err = "NULL output"; return CHAINDP_ERR_ARG; }
	if (R == 0 || n_c == 0) return CHAINDP_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = regs_per_read_buffers(ctx);
	if (rc) return rc;
	HIP_TRY(ctx, regs_grow(ctx, ctx->d_regs, ctx->regs_cap, (size_t)n_c * sizeof(chaindp_reg_t)));
	hipStream_t st = ctx->stream;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_rhash, hash, (size_t)R * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_rqlen, qlen, (size_t)R * 4, hipMemcpyHostToDevice, st));
	// sort keys go to the backtrack's 16-byte scratch (free once the chains are out), range stacks to its stack area
	HIP_TRY(ctx, chaindp::launch_gen_regs(st, R, ctx->bot.chains_off, ctx->bot.b_off, ctx->bot.u_out, ctx->bot.b_out, ctx->d_rhash, ctx->d_rqlen,
	                                      ctx->bot.w, ctx->bot.stacks, ctx->d_regs));
	HIP_TRY(ctx, hipMemcpyAsync(regs, ctx->d_regs, (size_t)n_c * sizeof(chaindp_reg_t), hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipStreamSynchronize(st));
	return CHAINDP_OK;
}

extern "C" int chaindp_est_err(chaindp_ctx_t *ctx, const int64_t *regs_off, chaindp_reg_t *regs, const int32_t *qlen,
                               const int32_t *ref_len, int32_t n_ref, const int64_t *mini_pos_off, const uint64_t *mini_pos,
                               int32_t *match_tot)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (ctx->bot_n_reads < 0 || ctx->bot_n_reads != ctx->n_reads || !ctx->bot.has) { ctx->err = "chaindp_est_err needs the chains of a chaindp_backtrack on this batch"; return CHAINDP_ERR_ARG; }
	const int64_t R = ctx->bot_n_reads;
	if (R == 0) return CHAINDP_OK;
	if (!regs_off || !qlen || (n_ref > 0 && !ref_len) || n_ref < 0) { ctx->err = "NULL argument"; return CHAINDP_ERR_ARG; }
	if (regs_off[0] != 0) { ctx->err = "regs_off must start at 0"; return CHAINDP_ERR_ARG; }
	for (int64_t r = 0; r < R; ++r) if (regs_off[r + 1] < regs_off[r]) { ctx->err = "regs_off must not decrease"; return CHAINDP_ERR_ARG; }
	const int64_t n_regs = regs_off[R];
	if (n_regs == 0) return CHAINDP_OK;
	if (!regs) { ctx->err = "NULL regs"; return CHAINDP_ERR_ARG; }
	const bool resident = mini_pos == nullptr && mini_pos_off == nullptr;
	if (resident && (!ctx->mp_resident || !ctx->d_mp_off)) { ctx->err = "no resident mini_pos: pass the arrays, or collect the seeds with chaindp_collect_seeds"; return CHAINDP_ERR_ARG; }
	if (!resident && !mini_pos_off) { ctx->err = "mini_pos without offsets"; return CHAINDP_ERR_ARG; }
	for (int64_t g = 0; g < n_regs; ++g) if (regs[g].cnt < 0 || regs[g].as < 0) { ctx->err = "hit with a negative count or offset"; return CHAINDP_ERR_ARG; }
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = regs_per_read_buffers(ctx);
	if (rc) return rc;
	hipStream_t st = ctx->stream;
	// every hit's anchors must lie inside its read's chain anchors: checked here, on the host's copy of the offsets
	{
		std::vector<int64_t> boff((size_t)R + 1);
		HIP_TRY(ctx, hipMemcpyAsync(boff.data(), ctx->bot.b_off, (size_t)(R + 1) * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(ctx, hipStreamSynchronize(st));
		for (int64_t r = 0; r < R; ++r)
			for (int64_t g = regs_off[r]; g < regs_off[r + 1]; ++g)
				if ((int64_t)regs[g].as + regs[g].cnt > boff[r + 1] - boff[r]) { ctx->err = "hit reaches beyond its read's chain anchors"; return CHAINDP_ERR_ARG; }
	}
	HIP_TRY(ctx, regs_grow(ctx, ctx->d_regs, ctx->regs_cap, (size_t)n_regs * sizeof(chaindp_reg_t)));
	HIP_TRY(ctx, regs_grow(ctx, ctx->d_reg_counts, ctx->reg_counts_cap, (size_t)n_regs * 8));
	HIP_TRY(ctx, regs_grow(ctx, ctx->d_ref_len, ctx->ref_len_cap, (size_t)(n_ref > 0 ? n_ref : 1) * 4));
	const int64_t *d_mpo = ctx->d_mp_off;
	const unsigned long long *d_mp = ctx->d_mini_pos;
	if (!resident) {
		const int64_t n_mp = mini_pos_off[R];
		if (n_mp < 0 || (n_mp > 0 && !mini_pos)) { ctx->err = "mini_pos announced but absent"; return CHAINDP_ERR_ARG; }
		HIP_TRY(ctx, regs_grow(ctx, ctx->d_mp_up, ctx->mp_up_cap, (size_t)(n_mp > 0 ? n_mp : 1) * 8));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_mp_off_up, mini_pos_off, (size_t)(R + 1) * 8, hipMemcpyHostToDevice, st));
		if (n_mp > 0) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_mp_up, mini_pos, (size_t)n_mp * 8, hipMemcpyHostToDevice, st));
		d_mpo = ctx->d_mp_off_up; d_mp = (const unsigned long long*)ctx->d_mp_up;
	}
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_regs_off, regs_off, (size_t)(R + 1) * 8, hipMemcpyHostToDevice, st));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_rqlen, qlen, (size_t)R * 4, hipMemcpyHostToDevice, st));
	if (n_ref > 0) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ref_len, ref_len, (size_t)n_ref * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_regs, regs, (size_t)n_regs * sizeof(chaindp_reg_t), hipMemcpyHostToDevice, st));
	HIP_TRY(ctx, chaindp::launch_est_err(st, R, n_regs, ctx->d_regs_off, ctx->bot.b_off, ctx->bot.b_out, ctx->d_rqlen, (const int32_t*)ctx->d_ref_len, n_ref,
	                                     d_mpo, d_mp, ctx->d_sum_k, ctx->d_regs, (int32_t*)ctx->d_reg_counts));
	HIP_TRY(ctx, hipMemcpyAsync(regs, ctx->d_regs, (size_t)n_regs * sizeof(chaindp_reg_t), hipMemcpyDeviceToHost, st));
	if (match_tot) HIP_TRY(ctx, hipMemcpyAsync(match_tot, ctx->d_reg_counts, (size_t)n_regs * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipStreamSynchronize(st));
	return CHAINDP_OK;
}

// test hook (not in the public header): units the two-per-wave kernel handed over to k_chain_units in the last run.  When that
// kernel declines the whole batch (long units: map-ont, dense repeats) the word on the device is the marker 0xffffffff, "every
// unit": reported as the batch's unit count
extern "C" int64_t chaindp_debug_leftover(chaindp_ctx_t *ctx)
{
	if (!ctx || !ctx->d_left_cnt) return -1;
	unsigned long long c = 0, cnt = 0;
	if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
	    hipMemcpy(&c, ctx->d_left_cnt, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess ||
	    hipMemcpy(&cnt, ctx->d_counters, sizeof(cnt), hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return (uint32_t)c == 0xffffffffu ? (int64_t)(uint32_t)cnt : (int64_t)(uint32_t)c;
}

// test hook (not in the public header): 1 lets k_chain_quad take the batches it can (one cost table, ordinary units), 0 (default) never
extern "C" int chaindp_debug_set_quad(chaindp_ctx_t *ctx, int on)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	ctx->use_quad = on != 0;
	return CHAINDP_OK;
}

// test hook (not in the public header): 1 if k_chain_quad took the last batch
extern "C" int chaindp_debug_quad_took(chaindp_ctx_t *ctx)
{
	if (!ctx || !ctx->d_left_cnt) return -1;
	unsigned long long r = 0;
	if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
	    hipMemcpy(&r, ctx->d_left_cnt + 3, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return (uint32_t)r != 0;
}

// test hook (not in the public header): what k_chain_twin hands over whatever the units look like -- 0 nothing extra, 1 every unit
// untouched, 2 every unit after its first 64-anchor tile (k_chain_units resumes behind it)
extern "C" int chaindp_debug_set_twin_handover(chaindp_ctx_t *ctx, int mode)
{
	if (!ctx || mode < 0 || mode > 2) return CHAINDP_ERR_ARG;
	ctx->twin_force_left = mode;
	return CHAINDP_OK;
}

// test hook (not in the public header): 0 keeps every unit in the launch that took it (the deep path of the small rings stays
// covered by the parity tests), 1 (default) hands long units whose scans keep reaching past the ring to k_chain_dense or, when the
// batch is dense all over, k_chain_dense1; 2 any unit with a few such scans, to k_chain_dense; 3 the same to k_chain_dense1; 4 the same
// to k_chain_dense16 (small test inputs reach every kernel)
extern "C" int chaindp_debug_set_deep_handover(chaindp_ctx_t *ctx, int on)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	ctx->deep_handover = on != 0;
	ctx->deep_eager = on >= 2;
	ctx->deep_route = on == 2 ? 1 : on == 3 ? 2 : on == 4 ? 3 : 0;
	return CHAINDP_OK;
}

// test hook (not in the public header): units k_chain_units handed over to k_chain_dense in the last run
extern "C" int64_t chaindp_debug_deep_units(chaindp_ctx_t *ctx)
{
	if (!ctx || !ctx->d_left_cnt) return -1;
	unsigned long long c = 0;
	if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
	    hipMemcpy(&c, ctx->d_left_cnt + 1, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return (int64_t)(uint32_t)c;
}

// test hook (not in the public header): fills one of the backtrack allocations (in allocation order) with a byte
extern "C" int chaindp_debug_poison(chaindp_ctx_t *ctx, int which, int byte, size_t bytes)
{
	if (!ctx || which < 0 || (size_t)which >= ctx->bot_allocs.size()) return CHAINDP_ERR_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemset(ctx->bot_allocs[which], byte, bytes));
	return CHAINDP_OK;
}

// test hook (not in the public header): copies one of the backtrack scratch arrays to the host
extern "C" int chaindp_debug_bottom(chaindp_ctx_t *ctx, int which, void *dst, size_t bytes)
{
	if (!ctx || !ctx->bot.has) return CHAINDP_ERR_ARG;
	const void *src = nullptr;
	switch (which) {
	case 0: src = ctx->bot.has; break;
	case 1: src = ctx->bot.owner; break;
	case 2: src = ctx->bot.skey; break;
	case 3: src = ctx->bot.ccnt; break;
	case 4: src = ctx->bot.key; break;
	case 5: src = ctx->bot.end_rec; break;
	case 6: src = ctx->bot.ends_off; break;
	default: return CHAINDP_ERR_ARG;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
	return CHAINDP_OK;
}

extern "C" int chaindp_run_full(chaindp_ctx_t *ctx, const chaindp_params_t *par)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	int rc = chaindp_run(ctx, par);
	if (rc) return rc;
	return compact_launch(ctx, par);
}

extern "C" int chaindp_compact_offsets(chaindp_ctx_t *ctx, const chaindp_params_t *par, int64_t *seeds_off)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	return compact_on_device(ctx, par, seeds_off);
}

extern "C" int chaindp_download_seeds(chaindp_ctx_t *ctx, int64_t first_seed, int64_t n_seeds, chaindp_seed_t *dst)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (first_seed < 0 || n_seeds < 0 || first_seed + n_seeds > ctx->n_seeds || (n_seeds > 0 && !dst)) {
		ctx->err = "seed range outside the last compaction";
		return CHAINDP_ERR_ARG;
	}
	if (n_seeds == 0) return CHAINDP_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(dst, (const chaindp_seed_t*)ctx->d_seeds + first_seed, (size_t)n_seeds * sizeof(chaindp_seed_t),
	                            hipMemcpyDeviceToHost, ctx->stream));
	return CHAINDP_OK;
}

extern "C" int chaindp_compact(chaindp_ctx_t *ctx, const chaindp_params_t *par, int64_t *seeds_off, chaindp_seed_t *seeds)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (ctx->total > 0 && !seeds) { ctx->err = "NULL output"; return CHAINDP_ERR_ARG; }
	int rc = compact_on_device(ctx, par, seeds_off);
	if (rc) return rc;
	if ((rc = chaindp_download_seeds(ctx, 0, ctx->n_seeds, seeds)) != CHAINDP_OK) return rc;
	return chaindp_sync(ctx);
}

// device array of n_reads host pointers (grown on demand)
static int stage_pointers(chaindp_ctx *ctx, const void *const *ptrs, int64_t n)
{
	if ((size_t)n > ctx->ptr_cap) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->d_ptrs) HIP_TRY(ctx, hipFree(ctx->d_ptrs));
		ctx->d_ptrs = nullptr; ctx->ptr_cap = 0;
		const size_t cap = (size_t)n + (size_t)n / 2 + 64;
		HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ptrs, cap * sizeof(void*)));
		ctx->ptr_cap = cap;
	}
	if (n) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ptrs, ptrs, (size_t)n * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
	return CHAINDP_OK;
}

extern "C" int chaindp_upload_gather_ex(chaindp_ctx_t *ctx, int64_t n_reads, const int64_t *off,
                                        const chaindp_anchor_t *const *read_anchors, const int32_t *n_segs_per_read, int pinned)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (!pinned) return chaindp_upload_gather(ctx, n_reads, off, read_anchors, n_segs_per_read);
	if (n_reads < 0 || !off || (n_reads > 0 && (off[0] != 0 || !read_anchors))) { ctx->err = "bad offsets"; return CHAINDP_ERR_ARG; }
	const int64_t total = n_reads > 0 ? off[n_reads] : 0;
	if (n_reads > ctx->cap_reads || total > ctx->cap_anchors) {
		ctx->err = "batch exceeds the capacity the context was created with";
		return CHAINDP_ERR_CAPACITY;
	}
	for (int64_t r = 0; r < n_reads; ++r)
		if (off[r + 1] < off[r] || (off[r + 1] > off[r] && !read_anchors[r])) { ctx->err = "bad read in gather list"; return CHAINDP_ERR_ARG; }
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_off, off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
	int rc = stage_pointers(ctx, (const void *const *)read_anchors, n_reads);
	if (rc) return rc;
	HIP_TRY(ctx, chaindp::launch_gather_reads(ctx->stream, n_reads, ctx->d_off, (const void *const *)ctx->d_ptrs, ctx->d_a));
	ctx->has_n_segs = n_segs_per_read != nullptr;
	if (n_segs_per_read && n_reads)
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_n_segs, n_segs_per_read, (size_t)n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // the host arrays (off, pointers) may go away after the call
	ctx->n_reads = n_reads; ctx->total = total; ctx->ran = false; ctx->bot_n_reads = -1; ctx->mp_resident = false;
	return CHAINDP_OK;
}

extern "C" int chaindp_scatter_seeds(chaindp_ctx_t *ctx, int64_t n_reads, chaindp_seed_t *const *dst)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (n_reads != ctx->n_reads || (n_reads > 0 && !dst) || !ctx->d_seeds) { ctx->err = "scatter does not match the last compaction"; return CHAINDP_ERR_ARG; }
	for (int64_t r = 0; r < n_reads; ++r) if ((uintptr_t)dst[r] & 15u) { ctx->err = "scatter destinations must be 16-byte aligned"; return CHAINDP_ERR_ARG; }
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = stage_pointers(ctx, (const void *const *)dst, n_reads);
	if (rc) return rc;
	HIP_TRY(ctx, chaindp::launch_scatter_seeds(ctx->stream, n_reads, ctx->d_seeds_off, (void *const *)ctx->d_ptrs, ctx->d_seeds));
	return CHAINDP_OK;
}

extern "C" int chaindp_upload_gather(chaindp_ctx_t *ctx, int64_t n_reads, const int64_t *off,
                                     const chaindp_anchor_t *const *read_anchors, const int32_t *n_segs_per_read)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (n_reads < 0 || !off || (n_reads > 0 && (off[0] != 0 || !read_anchors))) { ctx->err = "bad offsets"; return CHAINDP_ERR_ARG; }
	const int64_t total = n_reads > 0 ? off[n_reads] : 0;
	if (n_reads > ctx->cap_reads || total > ctx->cap_anchors) {
		ctx->err = "batch exceeds the capacity the context was created with";
		return CHAINDP_ERR_CAPACITY;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_off, off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
	for (int64_t r = 0; r < n_reads; ++r) {
		const int64_t n = off[r + 1] - off[r];
		if (n < 0 || (n > 0 && !read_anchors[r])) { ctx->err = "bad read in gather list"; return CHAINDP_ERR_ARG; }
		if (n) HIP_TRY(ctx, hipMemcpyAsync((chaindp_anchor_t*)ctx->d_a + off[r], read_anchors[r], (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
	}
	ctx->has_n_segs = n_segs_per_read != nullptr;
	if (n_segs_per_read && n_reads)
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_n_segs, n_segs_per_read, (size_t)n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ctx->n_reads = n_reads; ctx->total = total; ctx->ran = false; ctx->bot_n_reads = -1; ctx->mp_resident = false;
	return CHAINDP_OK;
}

extern "C" void *chaindp_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
	return p;
}

extern "C" void chaindp_host_free(void *p)
{
	if (p) (void)hipHostFree(p);
}

extern "C" int chaindp_get_kernel_ms(chaindp_ctx_t *ctx, double ms[4], int64_t launches[4], int reset)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	for (auto &es : ctx->pending) {
		HIP_TRY(ctx, hipEventSynchronize(es.e[es.n - 1]));
		for (int k = 0; k + 1 < es.n; ++k) {
			float t = 0;
			HIP_TRY(ctx, hipEventElapsedTime(&t, es.e[k], es.e[k + 1]));
			ctx->ms[es.slot0 + k] += (double)t;
			ctx->launches[es.slot0 + k] += 1;
		}
		for (int k = 0; k < es.n; ++k) (void)hipEventDestroy(es.e[k]);
	}
	ctx->pending.clear();
	for (int k = 0; k < 4; ++k) { if (ms) ms[k] = ctx->ms[k]; if (launches) launches[k] = ctx->launches[k]; }
	if (reset) for (int k = 0; k < 4; ++k) { ctx->ms[k] = 0; ctx->launches[k] = 0; }
	return CHAINDP_OK;
}

extern "C" int chaindp_get_stats(chaindp_ctx_t *ctx, int64_t st[4])
{
	if (!ctx || !st) return CHAINDP_ERR_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	unsigned long long c = 0;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	HIP_TRY(ctx, hipMemcpy(&c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
	ctx->stats[0] = (int64_t)(c & 0xffffffffull); ctx->stats[1] = (int64_t)(c >> 32);
	for (int k = 0; k < 4; ++k) st[k] = ctx->stats[k];
	return CHAINDP_OK;
}

// ---- seed collection on the GPU (chaindp_seed.hip)

struct chaindp_index {
	int device = -1;
	uint8_t *blob[4] = {nullptr, nullptr, nullptr, nullptr};
	size_t bytes[4] = {0, 0, 0, 0};
	int b_bits = 0;
};

extern "C" chaindp_index_t *chaindp_index_create(int device, const void *B, size_t nB, const void *H, size_t nH,
                                                 const void *V, size_t nV, const void *P, size_t nP)
{
	g_create_error.clear();
	if (!B || nB < 16 || !H || !V) { g_create_error = "chaindp_index_create: the image needs its B, H and V blobs"; return nullptr; }
	if (hipSetDevice(device) != hipSuccess) { g_create_error = "chaindp_index_create: no such HIP device (there is no CPU fallback)"; return nullptr; }
	chaindp_index *ix = new chaindp_index();
	ix->device = device;
	const void *src[4] = {B, H, V, P};
	const size_t nb[4] = {nB, nH, nV, nP};
	for (int k = 0; k < 4; ++k) {
		ix->bytes[k] = nb[k];
		const size_t alloc = (nb[k] + 63) & ~(size_t)63;                 // the kernels read whole 64-byte groups
		if (hipMalloc((void**)&ix->blob[k], alloc ? alloc : 64) != hipSuccess ||
		    hipMemset(ix->blob[k], 0, alloc ? alloc : 64) != hipSuccess ||
		    (nb[k] && hipMemcpy(ix->blob[k], src[k], nb[k], hipMemcpyHostToDevice) != hipSuccess)) {
			g_create_error = "chaindp_index_create: out of device memory";
			chaindp_index_destroy(ix);
			return nullptr;
		}
	}
	size_t entries = nB / 16;
	while ((size_t)2 << ix->b_bits <= entries) ++ix->b_bits;          // one 16-byte entry per bucket, 2^b buckets
	return ix;
}

extern "C" void chaindp_index_destroy(chaindp_index_t *ix)
{
	if (!ix) return;
	if (ix->device >= 0) (void)hipSetDevice(ix->device);
	for (int k = 0; k < 4; ++k) if (ix->blob[k]) (void)hipFree(ix->blob[k]);
	delete ix;
}

static int seed_reserve(chaindp_ctx *ctx, int64_t n_mini)
{
	if (!ctx->seed_ready) {
		// first use: all or nothing, as in compact_launch
		const size_t nr = (size_t)ctx->cap_reads;
		void **slot[7] = {(void**)&ctx->d_mini_off, (void**)&ctx->d_mp_off, (void**)&ctx->d_bid, (void**)&ctx->d_qlen, (void**)&ctx->d_rep_len,
		                  (void**)&ctx->seed.totals, (void**)&ctx->seed.stacks};
		const size_t stack_bytes = ((size_t)ctx->cap_anchors / 64 + 2 * nr + 4) * 12, tied_bytes = ((size_t)ctx->cap_anchors / 64 + nr + 8) * 4;
		const size_t sz[7] = {(nr + 1) * 8, (nr + 1) * 8, (nr + 1) * 4, (nr + 1) * 4, (nr + 1) * 4, 32, stack_bytes + tied_bytes};
		void *nb[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
		hipError_t e = hipSuccess;
		for (int k = 0; k < 7 && e == hipSuccess; ++k) e = hipMalloc(&nb[k], sz[k]);
		if (e != hipSuccess) {
			for (void *q : nb) if (q) (void)hipFree(q);
			ctx->err = std::string("seed collection buffers: ") + hipGetErrorString(e);
			return CHAINDP_ERR_HIP;
		}
		for (int k = 0; k < 7; ++k) *slot[k] = nb[k];
		ctx->seed.tied = (uint32_t*)((char*)ctx->seed.stacks + stack_bytes);
		ctx->seed_ready = true;
	}
	if (n_mini > ctx->seed_cap_mini) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		void **grow[] = {(void**)&ctx->seed.kept, (void**)&ctx->seed.used, (void**)&ctx->seed.src, (void**)&ctx->seed.mstate,
		                 (void**)&ctx->seed.tile_tmp, (void**)&ctx->d_mini, (void**)&ctx->d_mini_pos};
		for (void **g : grow) if (*g) { HIP_TRY(ctx, hipFree(*g)); *g = nullptr; }
		ctx->seed_cap_mini = 0;
		const size_t n = (size_t)n_mini + (size_t)n_mini / 4 + 1024;
		HIP_TRY(ctx, hipMalloc((void**)&ctx->seed.kept, n * 8));
		HIP_TRY(ctx, hipMalloc((void**)&ctx->seed.used, n * 8));
		HIP_TRY(ctx, hipMalloc((void**)&ctx->seed.src, n * 8));
		HIP_TRY(ctx, hipMalloc((void**)&ctx->seed.mstate, n * 8));
		HIP_TRY(ctx, hipMalloc((void**)&ctx->seed.tile_tmp, (n / 1024 + 2) * 8));
		HIP_TRY(ctx, hipMalloc((void**)&ctx->d_mini, n * 16));
		HIP_TRY(ctx, hipMalloc((void**)&ctx->d_mini_pos, n * 8));
		ctx->seed_cap_mini = (int64_t)n;
	}
	return CHAINDP_OK;
}

// mini: all minimizers contiguous (read_mini == NULL), or read_mini[r] = read r's minimizers in pinned host memory
static int collect_seeds_impl(chaindp_ctx *ctx, const chaindp_index_t *ix, int flag, int max_occ, int64_t n_reads,
                              const int64_t *mini_off, const chaindp_anchor_t *mini, const chaindp_anchor_t *const *read_mini,
                              const uint32_t *bid, const int32_t *qlen,
                              const int32_t *n_segs_per_read, int64_t *off, int32_t *rep_len, int64_t *mini_pos_off)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (!ix || ix->device != ctx->device) { ctx->err = "index image missing or on another device"; return CHAINDP_ERR_ARG; }
	if (n_reads < 0 || !mini_off || (n_reads > 0 && (mini_off[0] != 0 || !bid || !qlen))) { ctx->err = "bad minimizer offsets"; return CHAINDP_ERR_ARG; }
	const int64_t n_mini = n_reads > 0 ? mini_off[n_reads] : 0;
	if (n_mini < 0 || (n_mini > 0 && !mini && !read_mini)) { ctx->err = "bad minimizers"; return CHAINDP_ERR_ARG; }
	if (n_reads > ctx->cap_reads) { ctx->err = "batch exceeds the capacity the context was created with"; return CHAINDP_ERR_CAPACITY; }
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = seed_reserve(ctx, n_mini);
	if (rc) return rc;
	hipStream_t st = ctx->stream;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_mini_off, mini_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
	if (n_mini && read_mini) {                                     // one kernel pulls every read's minimizers out of its pinned buffer
		rc = stage_pointers(ctx, (const void *const *)read_mini, n_reads);
		if (rc) return rc;
		HIP_TRY(ctx, chaindp::launch_gather_reads(st, n_reads, ctx->d_mini_off, (const void *const *)ctx->d_ptrs, ctx->d_mini));
	} else if (n_mini) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_mini, mini, (size_t)n_mini * 16, hipMemcpyHostToDevice, st));
	if (n_reads) {
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bid, bid, (size_t)n_reads * 4, hipMemcpyHostToDevice, st));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_qlen, qlen, (size_t)n_reads * 4, hipMemcpyHostToDevice, st));
	}
	ctx->has_n_segs = n_segs_per_read != nullptr;
	if (n_segs_per_read && n_reads) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_n_segs, n_segs_per_read, (size_t)n_reads * 4, hipMemcpyHostToDevice, st));
	chaindp::SeedIndex dix;
	dix.B = ix->blob[0]; dix.H = ix->blob[1]; dix.V = ix->blob[2]; dix.P = ix->blob[3];
	dix.nB = ix->bytes[0]; dix.nH = ix->bytes[1]; dix.nV = ix->bytes[2]; dix.nP = ix->bytes[3];
	dix.b_bits = ix->b_bits;
	HIP_TRY(ctx, chaindp::launch_seed_collect(st, dix, flag, max_occ, n_reads, n_mini, ctx->d_mini_off, ctx->d_mini, ctx->d_bid, ctx->seed,
	                                          ctx->d_off, ctx->d_mp_off, ctx->d_rep_len));
	unsigned long long totals[2] = {0, 0};
	HIP_TRY(ctx, hipMemcpyAsync(totals, ctx->seed.totals, 16, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipStreamSynchronize(st));
	if ((int64_t)totals[0] > ctx->cap_anchors) {
		ctx->n_reads = 0; ctx->total = 0; ctx->ran = false; ctx->bot_n_reads = -1; ctx->mp_resident = false;
		ctx->err = "the batch's seeds exceed the anchor capacity the context was created with";
		return CHAINDP_ERR_CAPACITY;
	}
	// unsorted anchors go to the new_seed[] buffer (free at this point of a batch), the sort writes d_a
	if (!ctx->d_seeds) HIP_TRY(ctx, hipMalloc(&ctx->d_seeds, (size_t)ctx->cap_anchors * sizeof(chaindp_seed_t) + 16));
	if (ctx->seed_max_n < 0) {
		int lds_limit = 0;
		HIP_TRY(ctx, hipDeviceGetAttribute(&lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device));
		int m = 8192, m2 = 65024, cap = (lds_limit - 8192) & ~15;
		// test switches: smaller limits send ordinary reads down the paths made for very large ones
		if (const char *v = getenv("CHAINDP_SEED_MAX_N")) { m = atoi(v) < m ? atoi(v) : m; if (const char *c = strchr(v, ',')) m2 = atoi(c + 1); }
		if (const char *v = getenv("CHAINDP_SEED_LAB_CAP")) cap = atoi(v) < cap ? atoi(v) & ~15 : cap;
		if (m < 64) m = 64;
		if (cap < 256) cap = 256;
		while (m > 0 && chaindp::seed_sort_lds_bytes(m, 32, 8) > (size_t)lds_limit) m -= 512;
		while (m2 > m && chaindp::seed_sort_lds_bytes(m2, 4, 2) > (size_t)lds_limit) m2 -= 64;
		ctx->seed_lab_cap = cap;
		ctx->seed_max_n = m; ctx->seed_max_n2 = m2;
	}
	HIP_TRY(ctx, chaindp::launch_seed_expand_sort(st, dix, flag, n_reads, n_mini, ctx->d_mini_off, ctx->d_mini, ctx->d_bid, ctx->d_qlen, ctx->seed,
	                                              ctx->d_seeds, ctx->d_a, ctx->d_off, ctx->d_mini_pos, ctx->seed_max_n, ctx->seed_max_n2,
	                                              ctx->seed_lab_cap, (int64_t)totals[0]));
	if (off) HIP_TRY(ctx, hipMemcpyAsync(off, ctx->d_off, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, st));
	if (mini_pos_off) HIP_TRY(ctx, hipMemcpyAsync(mini_pos_off, ctx->d_mp_off, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, st));
	if (rep_len && n_reads) HIP_TRY(ctx, hipMemcpyAsync(rep_len, ctx->d_rep_len, (size_t)n_reads * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipStreamSynchronize(st));
	ctx->n_reads = n_reads; ctx->total = (int64_t)totals[0]; ctx->n_mini_pos = (int64_t)totals[1]; ctx->ran = false; ctx->bot_n_reads = -1; ctx->mp_resident = true;
	return CHAINDP_OK;
}

extern "C" int chaindp_collect_seeds(chaindp_ctx_t *ctx, const chaindp_index_t *ix, int flag, int max_occ, int64_t n_reads,
                                     const int64_t *mini_off, const chaindp_anchor_t *mini, const uint32_t *bid, const int32_t *qlen,
                                     const int32_t *n_segs_per_read, int64_t *off, int32_t *rep_len, int64_t *mini_pos_off)
{
	return collect_seeds_impl(ctx, ix, flag, max_occ, n_reads, mini_off, mini, nullptr, bid, qlen, n_segs_per_read, off, rep_len, mini_pos_off);
}

extern "C" int chaindp_collect_seeds_gather(chaindp_ctx_t *ctx, const chaindp_index_t *ix, int flag, int max_occ, int64_t n_reads,
                                            const int64_t *mini_off, const chaindp_anchor_t *const *read_mini, const uint32_t *bid,
                                            const int32_t *qlen, const int32_t *n_segs_per_read, int64_t *off, int32_t *rep_len,
                                            int64_t *mini_pos_off)
{
	if (ctx && n_reads > 0 && !read_mini) { ctx->err = "bad minimizers"; return CHAINDP_ERR_ARG; }
	return collect_seeds_impl(ctx, ix, flag, max_occ, n_reads, mini_off, nullptr, read_mini, bid, qlen, n_segs_per_read, off, rep_len, mini_pos_off);
}

extern "C" int chaindp_map_batch(chaindp_ctx_t *ctx, const chaindp_index_t *ix, int flag, int max_occ, const chaindp_params_t *par, int min_cnt,
                                 int64_t n_reads, const int64_t *mini_off, const chaindp_anchor_t *mini, const uint32_t *bid, const int32_t *qlen,
                                 const uint32_t *hash, int64_t *regs_off, chaindp_reg_t *regs, int64_t regs_cap, int32_t *rep_len, int64_t *n_anchors)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	int rc = check_params(ctx, par);
	if (rc) return rc;
	if (!regs_off || regs_cap < 0 || (regs_cap > 0 && !regs) || (n_reads > 0 && !hash)) { ctx->err = "NULL output or hash"; return CHAINDP_ERR_ARG; }
	// seeds (resident), DP + compaction, chains, hits: every stage reads what the one before left in HBM
	if ((rc = collect_seeds_impl(ctx, ix, flag, max_occ, n_reads, mini_off, mini, nullptr, bid, qlen, nullptr, nullptr, rep_len, nullptr)) != CHAINDP_OK) return rc;
	if (n_anchors) *n_anchors = ctx->total;
	if ((rc = chaindp_run_full(ctx, par)) != CHAINDP_OK) return rc;
	std::vector<int64_t> b_off((size_t)(n_reads > 0 ? n_reads + 1 : 1));
	if ((rc = chaindp_backtrack(ctx, par, min_cnt, regs_off, nullptr, b_off.data(), nullptr)) != CHAINDP_OK) return rc;
	const int64_t n_c = n_reads > 0 ? regs_off[n_reads] : 0;
	if (n_c > regs_cap) { ctx->err = "more hits than regs has room for (regs_off is valid; chaindp_gen_regs with a larger buffer returns them)"; return CHAINDP_ERR_CAPACITY; }
	return chaindp_gen_regs(ctx, hash, qlen, regs);
}

extern "C" int chaindp_scatter_mini_pos(chaindp_ctx_t *ctx, int64_t n_reads, uint64_t *const *dst)
{
	if (!ctx) return CHAINDP_ERR_ARG;
	if (n_reads != ctx->n_reads || (n_reads > 0 && !dst) || !ctx->d_mp_off) { ctx->err = "scatter does not match the last seed collection"; return CHAINDP_ERR_ARG; }
	for (int64_t r = 0; r < n_reads; ++r) if ((uintptr_t)dst[r] & 15u) { ctx->err = "scatter destinations must be 16-byte aligned"; return CHAINDP_ERR_ARG; }
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = stage_pointers(ctx, (const void *const *)dst, n_reads);
	if (rc) return rc;
	HIP_TRY(ctx, chaindp::launch_scatter_words(ctx->stream, n_reads, ctx->d_mp_off, (void *const *)ctx->d_ptrs, ctx->d_mini_pos));
	return CHAINDP_OK;
}

extern "C" int chaindp_download_mini_pos(chaindp_ctx_t *ctx, uint64_t *mini_pos)
{
	if (!ctx || (ctx->n_mini_pos > 0 && !mini_pos)) return CHAINDP_ERR_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (ctx->n_mini_pos) HIP_TRY(ctx, hipMemcpyAsync(mini_pos, ctx->d_mini_pos, (size_t)ctx->n_mini_pos * 8, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CHAINDP_OK;
}

extern "C" int chaindp_download_anchors(chaindp_ctx_t *ctx, chaindp_anchor_t *a)
{
	if (!ctx || (ctx->total > 0 && !a)) return CHAINDP_ERR_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (ctx->total) HIP_TRY(ctx, hipMemcpyAsync(a, ctx->d_a, (size_t)ctx->total * 16, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CHAINDP_OK;
}

// ---- streaming pipeline (include/chaindp.h): depth contexts, each with its own stream, round robin

struct PipeSlot {
	chaindp_ctx *ctx = nullptr;
	int64_t *h_seeds_off = nullptr;          // pinned
	chaindp_seed_t *h_seeds = nullptr;       // pinned
	unsigned long long *h_n_seeds = nullptr; // pinned
	hipEvent_t done = nullptr;               // kernels + small downloads of the batch
	hipEvent_t up = nullptr;                 // the batch's upload
	int64_t tag = 0, n_reads = 0, total = 0;
	int state = 0;                           // 0 free, 1 in flight, 2 waited (results in use)
};

struct chaindp_pipe {
	int device = -1, depth = 0;
	hipStream_t s_up = nullptr, s_down = nullptr;   // one stream per copy direction, shared by the slots: uploads and downloads of
	                                         // different batches then run on different DMA engines, at the same time
	std::vector<PipeSlot> slots;
	int head = 0, tail = 0, inflight = 0;    // tail: oldest submitted, head: next to submit
	std::string err;
};

extern "C" const char *chaindp_pipe_last_error(const chaindp_pipe_t *pipe)
{
	return pipe ? pipe->err.c_str() : g_create_error.c_str();
}

extern "C" void chaindp_pipe_destroy(chaindp_pipe_t *pipe)
{
	if (!pipe) return;
	if (pipe->device >= 0) (void)hipSetDevice(pipe->device);
	for (auto &sl : pipe->slots) {
		if (sl.ctx && sl.ctx->stream) (void)hipStreamSynchronize(sl.ctx->stream);
		if (sl.done) (void)hipEventDestroy(sl.done);
		if (sl.up) (void)hipEventDestroy(sl.up);
		if (sl.h_seeds_off) (void)hipHostFree(sl.h_seeds_off);
		if (sl.h_seeds) (void)hipHostFree(sl.h_seeds);
		if (sl.h_n_seeds) (void)hipHostFree(sl.h_n_seeds);
		if (sl.ctx) chaindp_destroy(sl.ctx);
	}
	if (pipe->s_up) { (void)hipStreamSynchronize(pipe->s_up); (void)hipStreamDestroy(pipe->s_up); }
	if (pipe->s_down) { (void)hipStreamSynchronize(pipe->s_down); (void)hipStreamDestroy(pipe->s_down); }
	delete pipe;
}

extern "C" chaindp_pipe_t *chaindp_pipe_create(int device, int depth, int64_t max_anchors, int64_t max_reads)
{
	g_create_error.clear();
	if (depth < 1 || depth > 8) { g_create_error = "chaindp_pipe_create: depth must be 1..8"; return nullptr; }
	chaindp_pipe *pipe = new chaindp_pipe();
	pipe->device = device; pipe->depth = depth;
	pipe->slots.resize((size_t)depth);
	if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&pipe->s_up, hipStreamNonBlocking) != hipSuccess ||
	    hipStreamCreateWithFlags(&pipe->s_down, hipStreamNonBlocking) != hipSuccess) {
		g_create_error = "chaindp_pipe_create: no usable HIP device (there is no CPU fallback)";
		chaindp_pipe_destroy(pipe);
		return nullptr;
	}
	for (auto &sl : pipe->slots) {
		sl.ctx = chaindp_create(device, max_anchors, max_reads);
		if (!sl.ctx) { chaindp_pipe_destroy(pipe); return nullptr; }
		const size_t na = (size_t)sl.ctx->cap_anchors, nr = (size_t)sl.ctx->cap_reads;
		hipError_t e = hipHostMalloc((void**)&sl.h_seeds_off, (nr + 1) * 8, hipHostMallocDefault);
		if (e == hipSuccess) e = hipHostMalloc((void**)&sl.h_seeds, na * sizeof(chaindp_seed_t) + 16, hipHostMallocDefault);
		if (e == hipSuccess) e = hipHostMalloc((void**)&sl.h_n_seeds, 64, hipHostMallocDefault);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.up, hipEventDisableTiming);
		if (e != hipSuccess) {
			g_create_error = std::string("chaindp_pipe_create: ") + hipGetErrorString(e);
			chaindp_pipe_destroy(pipe);
			return nullptr;
		}
	}
	return pipe;
}

#define PIPE_TRY(pipe, call)                                                                       \
	do {                                                                                           \
		hipError_t e_ = (call);                                                                    \
		if (e_ != hipSuccess) {                                                                    \
			(pipe)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
			return CHAINDP_ERR_HIP;                                                                \
		}                                                                                          \
	} while (0)

extern "C" int chaindp_pipe_submit(chaindp_pipe_t *pipe, const chaindp_params_t *par, int64_t n_reads, const int64_t *off,
                                   const chaindp_anchor_t *a, const int32_t *n_segs_per_read, int64_t tag)
{
	if (!pipe) return CHAINDP_ERR_ARG;
	if (pipe->inflight == pipe->depth) { pipe->err = "every slot of the pipe is in flight: wait for the oldest batch first"; return CHAINDP_ERR_BUSY; }
	PipeSlot &sl = pipe->slots[(size_t)pipe->head];
	chaindp_ctx *ctx = sl.ctx;
	int rc = check_params(ctx, par);
	if (rc) { pipe->err = ctx->err; return rc; }
	if (n_reads < 0 || !off || (n_reads > 0 && off[0] != 0)) { pipe->err = "bad offsets"; return CHAINDP_ERR_ARG; }
	const int64_t total = n_reads > 0 ? off[n_reads] : 0;
	if (total < 0 || (total > 0 && !a)) { pipe->err = "bad anchors"; return CHAINDP_ERR_ARG; }
	if (n_reads > ctx->cap_reads || total > ctx->cap_anchors) { pipe->err = "batch exceeds the capacity the pipe was created with"; return CHAINDP_ERR_CAPACITY; }
	PIPE_TRY(pipe, hipSetDevice(pipe->device));
	hipStream_t st = ctx->stream;
	// upload on the pipe's upload stream (the slot's previous batch has been waited for, so its buffers are free); the
	// slot's own stream takes over for the kernels once the upload is in
	PIPE_TRY(pipe, hipMemcpyAsync(ctx->d_off, off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, pipe->s_up));
	if (total) PIPE_TRY(pipe, hipMemcpyAsync(ctx->d_a, a, (size_t)total * 16, hipMemcpyHostToDevice, pipe->s_up));
	ctx->has_n_segs = n_segs_per_read != nullptr;
	if (n_segs_per_read && n_reads) PIPE_TRY(pipe, hipMemcpyAsync(ctx->d_n_segs, n_segs_per_read, (size_t)n_reads * 4, hipMemcpyHostToDevice, pipe->s_up));
	PIPE_TRY(pipe, hipEventRecord(sl.up, pipe->s_up));
	PIPE_TRY(pipe, hipStreamWaitEvent(st, sl.up, 0));
	ctx->n_reads = n_reads; ctx->total = total; ctx->ran = false; ctx->bot_n_reads = -1; ctx->mp_resident = false;
	rc = chaindp_run_full(ctx, par);
	if (rc) { pipe->err = ctx->err; return rc; }
	PIPE_TRY(pipe, hipMemcpyAsync(sl.h_seeds_off, ctx->d_seeds_off, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, st));
	PIPE_TRY(pipe, hipMemcpyAsync(sl.h_n_seeds, ctx->cmp.n_seeds, 8, hipMemcpyDeviceToHost, st));
	PIPE_TRY(pipe, hipEventRecord(sl.done, st));
	sl.tag = tag; sl.n_reads = n_reads; sl.total = total; sl.state = 1;
	pipe->head = (pipe->head + 1) % pipe->depth;
	++pipe->inflight;
	return CHAINDP_OK;
}

extern "C" int chaindp_pipe_wait(chaindp_pipe_t *pipe, chaindp_pipe_result_t *res)
{
	if (!pipe || !res) return CHAINDP_ERR_ARG;
	PipeSlot &sl = pipe->slots[(size_t)pipe->tail];
	if (pipe->inflight == 0 || sl.state != 1) { pipe->err = sl.state == 2 ? "release the batch waited for first" : "nothing in flight"; return CHAINDP_ERR_BUSY; }
	PIPE_TRY(pipe, hipSetDevice(pipe->device));
	PIPE_TRY(pipe, hipEventSynchronize(sl.done));
	// the record count is known now: download exactly the batch's new_seed[] (the other slots' uploads and kernels go on)
	const int64_t m = sl.total > 0 && sl.n_reads > 0 ? (int64_t)(uint32_t)*sl.h_n_seeds : 0;
	sl.ctx->n_seeds = m;
	if (m > 0) {
		// a few workgroups are enough to fill the link and leave the shader array to the other slots' kernels
		static const int copy_blocks = getenv("CHAINDP_PIPE_COPY_BLOCKS") ? atoi(getenv("CHAINDP_PIPE_COPY_BLOCKS")) : 64;
		if (copy_blocks > 0) PIPE_TRY(pipe, chaindp::launch_copy_out(pipe->s_down, sl.h_seeds, sl.ctx->d_seeds, (size_t)m * sizeof(chaindp_seed_t), copy_blocks));
		else PIPE_TRY(pipe, hipMemcpyAsync(sl.h_seeds, sl.ctx->d_seeds, (size_t)m * sizeof(chaindp_seed_t), hipMemcpyDeviceToHost, pipe->s_down));
		PIPE_TRY(pipe, hipStreamSynchronize(pipe->s_down));
	}
	if (sl.n_reads == 0) sl.h_seeds_off[0] = 0;
	res->tag = sl.tag; res->n_reads = sl.n_reads; res->n_anchors = sl.total; res->n_seeds = m;
	res->seeds_off = sl.h_seeds_off; res->seeds = sl.h_seeds;
	sl.state = 2;
	return CHAINDP_OK;
}

extern "C" int chaindp_pipe_release(chaindp_pipe_t *pipe)
{
	if (!pipe) return CHAINDP_ERR_ARG;
	PipeSlot &sl = pipe->slots[(size_t)pipe->tail];
	if (sl.state != 2) { pipe->err = "no waited batch to release"; return CHAINDP_ERR_ARG; }
	sl.state = 0;
	pipe->tail = (pipe->tail + 1) % pipe->depth;
	--pipe->inflight;
	return CHAINDP_OK;
}
