// fpga_shim.cpp -- the reference's accelerator driver ABI (fpga.h:37-62) served by MI355X GPUs.
// See include/chaindp_fpga.h for the contract and the reference call sites of each entry point.
//
// Structure: producers (the reference's worker threads, map.c:439-444) obtain pinned packet buffers
// and submit them; one service thread per GPU drains the submit queue, merges up to
// max_packets_per_batch packets into ONE device batch (reads grouped by their (gap_ref, gap_qry)
// pair, which is uniform in practice), pulls every read's anchors straight out of the pinned packets
// with one gather kernel, runs prepass + chain DP + compaction, and has one scatter kernel write every
// read's new_seed[] straight into its slot of a pinned result packet (no host-side bulk copies); the single consumer (recv_task_thread, fpga_chaindp.c:228) blocks
// in fpga_get_retbuf.  Results may return in any order (map.c:930,946 match by read_id).
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>
#include "../../include/chaindp_fpga.h"

static_assert(sizeof(chaindp_pkt_hdr_t) == 64, "chaindp_sndhdr_t must be 64 bytes (reference main.c:296-302)");
static_assert(sizeof(chaindp_pkt_task_t) == 64, "collect_task_t must be 64 bytes");
static_assert(sizeof(chaindp_pkt_result_t) == 64, "collect_result_t must be 64 bytes");
static_assert(sizeof(chaindp_seed_t) == 24, "struct new_seed must be 24 bytes");

namespace {

// Pinned buffers in power-of-two size classes, recycled.  Every buffer is either handed out (in_use_) or in a free list,
// never both: a second release of the same pointer is refused instead of putting it into the free list twice.  Buffers are
// carved out of 64 MiB slabs: one hipHostMalloc per packet buffer (a millisecond each, thousands until enough of them
// circulate) made the first seconds of a run several times slower than its steady state.
class PinnedPool {
public:
	void *get(size_t bytes)
	{
		size_t cap = 1 << 16;
		while (cap < bytes) cap <<= 1;
		std::lock_guard<std::mutex> g(mu_);
		auto &fl = free_[cap];
		void *p = nullptr;
		if (!fl.empty()) { p = fl.back(); fl.pop_back(); }
		else {
			if (slab_left_ < cap) {
				const size_t sz = cap > kSlab ? cap : kSlab;
				void *sl = nullptr;
				if (hipHostMalloc(&sl, sz, hipHostMallocDefault) != hipSuccess) return nullptr;
				slabs_.push_back(sl);
				slab_cur_ = (char*)sl; slab_left_ = sz;              // (what was left of the previous slab is given up: less than one buffer)
			}
			p = slab_cur_; slab_cur_ += cap; slab_left_ -= cap;
		}
		cap_of_[p] = cap;
		in_use_[p] = bytes;
		return p;
	}
	// returns the size the buffer was handed out for, 0 if p is not a buffer in use
	size_t put(void *p)
	{
		std::lock_guard<std::mutex> g(mu_);
		auto it = in_use_.find(p);
		if (it == in_use_.end()) return 0;
		const size_t granted = it->second ? it->second : 1;
		in_use_.erase(it);
		free_[cap_of_[p]].push_back(p);
		return granted;
	}
	// size the buffer was handed out for, 0 if it is not in use
	size_t granted(void *p)
	{
		std::lock_guard<std::mutex> g(mu_);
		auto it = in_use_.find(p);
		return it == in_use_.end() ? 0 : (it->second ? it->second : 1);
	}
	bool owns(void *p) { return granted(p) != 0; }
	void destroy()
	{
		std::lock_guard<std::mutex> g(mu_);
		for (void *sl : slabs_) (void)hipHostFree(sl);
		slabs_.clear(); slab_cur_ = nullptr; slab_left_ = 0;
		cap_of_.clear(); free_.clear(); in_use_.clear();
	}
private:
	static constexpr size_t kSlab = 64u << 20;
	std::mutex mu_;
	std::map<size_t, std::vector<void*>> free_;
	std::unordered_map<void*, size_t> cap_of_, in_use_;
	std::vector<void*> slabs_;
	char *slab_cur_ = nullptr;
	size_t slab_left_ = 0;
};

struct Submitted { void *buf; uint32_t size; size_t granted; };
struct Result { void *buf; int size; };


// The index image as the reference streams it (index.c:102-119: blobs B, H, V, P as types 4..7, in chunks, in this order
// for every index part).  The shim keeps the bytes until the first service context of a GPU copies them to HBM
// (chaindp_index_create); the lookup itself runs on the GPU (chaindp_seed.hip).  An image is only used once it is SEALED: the
// first call that is not fpga_load_index (fpga_set_params, main.c:243, or a submit) closes it -- V arrives in 1 GB chunks and
// P last, so "has some of each blob" does not mean complete.  A sealed image is immutable; the next chunk starts a new one.
class IndexImage {
public:
	void append(int type, const void *data, size_t bytes)
	{
		std::vector<uint8_t> *dst = type == 4 ? &blob_[0] : type == 5 ? &blob_[1] : type == 6 ? &blob_[2] : type == 7 ? &blob_[3] : nullptr;
		if (!dst || !data || bytes == 0) return;
		const uint8_t *p = (const uint8_t*)data;
		dst->insert(dst->end(), p, p + bytes);
	}
	bool usable() const { return !blob_[0].empty() && !blob_[1].empty() && !blob_[2].empty(); }
	const std::vector<uint8_t> &blob(int k) const { return blob_[k]; }   // 0..3 = B, H, V, P
private:
	std::vector<uint8_t> blob_[4];
};

// A device copy of one sealed index image.  Held by reference count: every batch keeps its copy alive until its kernels are
// done, so an image that is replaced in mid-stream is freed by the last batch that looked seeds up in it, never under one.
struct DevIndex {
	chaindp_index_t *idx;
	explicit DevIndex(chaindp_index_t *i) : idx(i) {}
	~DevIndex() { if (idx) chaindp_index_destroy(idx); }
	DevIndex(const DevIndex&) = delete;
	DevIndex &operator=(const DevIndex&) = delete;
};

// what the service contexts of one GPU share: the device copy of the index image (one per GPU, not one per context)
struct GpuShare {
	std::mutex mu;                      // serialises the (multi-GB) upload; never taken together with the service mutex
	std::shared_ptr<DevIndex> idx;      // the copy of the NEWEST image a context of this GPU has seen
	uint64_t gen = 0;                   // generation of the image idx was made from (only ever grows)
	int64_t batches = 0, anchors = 0;   // work done on this GPU (under Service::mu)
};

struct Service {
	bool up = false;
	int n_gpus_cfg = 0, n_groups_cfg = 0, max_packets = 8192, services_per_gpu = 2;
	std::vector<int> group_device;               // service group -> GPU (one group per GPU unless chaindp_fpga_configure_groups says otherwise)
	int64_t cap_anchors = 32ll << 20, cap_reads = 1 << 19;   // device batch capacity per context (512 MiB of anchors)
	unsigned long max_inflight = 1ul << 30;
	// fpga_set_params (main.c:243)
	int bw = 500, is_cdna = 0, max_skip = 25, min_sc = 40, flag = 0, max_occ = 0;
	std::shared_ptr<IndexImage> building;        // fpga_load_index (main.c:201-204) appends here
	std::shared_ptr<const IndexImage> sealed;    // the image minimizer packets are looked up in
	uint64_t sealed_gen = 0;
	std::vector<std::unique_ptr<GpuShare>> gpus;
	PinnedPool pool;
	std::mutex mu;
	std::condition_variable cv_submit, cv_result;
	std::deque<Submitted> submit_q;
	std::deque<Result> result_q;
	unsigned long inflight_bytes = 0;
	bool stopping = false, exit_block = false, warned_capacity = false;
	std::vector<std::thread> workers;
	int64_t stats[5] = {0, 0, 0, 0, 0};
	uint32_t next_magic = 0;
};

Service g;

struct ReadRef {
	const chaindp_pkt_task_t *task;
	const chaindp_anchor_t *anchors;   // anchor packets: the payload; minimizer packets: filled in by the seed collection
	const chaindp_anchor_t *mini;      // minimizer packets (type 3): the payload
	int64_t n_anchors;
	int pkt, idx;          // position in the batch's packet list / within the packet
	bool on_device;        // false: answered with err_flag = 1
	int64_t batch_read;    // index in the device batch
	int rep_len;           // minimizer packets: collect_result_t::rep_len / n_minipos and the mini_pos[] payload
	int64_t n_minipos;
	std::vector<uint64_t> mini_pos;    // only when the payload has to be staged on the host (several groups in one batch)
};

void fail_hard(const char *what)
{
	// the reference's convention for driver failures is exit(1) (fpga_chaindp.c:105-109,246-249)
	fprintf(stderr, "[chaindp-fpga] fatal: %s\n", what);
	exit(1);
}

// must be called with g.mu held: the image being received becomes the one packets are looked up in
void seal_index_locked()
{
	if (g.building && g.building->usable()) {
		g.sealed = g.building;
		g.building.reset();
		++g.sealed_gen;
	}
}

void service_loop(int group)
{
	const int device = g.group_device[(size_t)group];
	const int64_t cap_anchors = g.cap_anchors, cap_reads = g.cap_reads;
	chaindp_ctx_t *ctx = chaindp_create(device, cap_anchors, cap_reads);
	if (!ctx) { fprintf(stderr, "[chaindp-fpga] %s\n", chaindp_last_error(nullptr)); fail_hard("cannot create a device context"); }
	GpuShare &share = *g.gpus[(size_t)group];
	std::vector<Submitted> pk;
	std::vector<ReadRef> reads;
	for (;;) {
		pk.clear(); reads.clear();
		int bw, is_cdna, max_skip, min_sc, sflag, max_occ;
		std::shared_ptr<const IndexImage> image;
		uint64_t image_gen;
		{
			std::unique_lock<std::mutex> lk(g.mu);
			g.cv_submit.wait(lk, [] { return g.stopping || !g.submit_q.empty(); });
			if (g.submit_q.empty()) break;          // stopping and drained
			// a device batch: packets until the anchor capacity (anchor packets: their payload; minimizer packets: a guess of four
			// hits per minimizer, checked for real after the lookup) -- batches are sized by work, not by packet count
			int64_t anchors_est = 0;
			while (!g.submit_q.empty() && (int)pk.size() < g.max_packets) {
				const Submitted sb = g.submit_q.front();
				const int64_t est = (int64_t)(sb.size / 16) * (((const chaindp_pkt_hdr_t*)sb.buf)->type == CHAINDP_PKT_MINIMIZERS ? 4 : 1);
				if (!pk.empty() && anchors_est + est > cap_anchors) break;
				anchors_est += est;
				pk.push_back(sb); g.submit_q.pop_front();
			}
			bw = g.bw; is_cdna = g.is_cdna; max_skip = g.max_skip; min_sc = g.min_sc;
			sflag = g.flag; max_occ = g.max_occ; image = g.sealed; image_gen = g.sealed_gen;
		}
		// this GPU's copy of the index image: made by whichever of its contexts gets here first, outside the service mutex
		// (producers and the receiver keep going during the multi-GB upload)
		// The batch holds a reference until it is done (index_ref is dropped at the end of this iteration): the other context of the
		// GPU may replace the shared copy meanwhile.  A context that still holds an OLDER image than the shared copy (it took its
		// packets just before the new image was sealed) makes a copy of its own for this one batch and leaves the shared one alone.
		std::shared_ptr<DevIndex> index_ref;
		if (image) {
			auto upload = [&]() {
				chaindp_index_t *ix = chaindp_index_create(device, image->blob(0).data(), image->blob(0).size(), image->blob(1).data(), image->blob(1).size(),
				                                          image->blob(2).data(), image->blob(2).size(), image->blob(3).data(), image->blob(3).size());
				if (!ix) { fprintf(stderr, "[chaindp-fpga] %s\n", chaindp_last_error(nullptr)); fail_hard("cannot load the index image onto the device"); }
				return std::make_shared<DevIndex>(ix);
			};
			std::lock_guard<std::mutex> gl(share.mu);
			if (!share.idx || image_gen > share.gen) { share.idx = upload(); share.gen = image_gen; }
			index_ref = share.gen == image_gen ? share.idx : upload();
		}
		chaindp_index_t *const dev_index = index_ref ? index_ref->idx : nullptr;
		const bool have_index = dev_index != nullptr;
		// ---- parse (map.c:484-568 walks the packet the same way; fpga_writebuf_submit has checked that the headers fit)
		for (size_t k = 0; k < pk.size(); ++k) {
			const char *base = (const char*)pk[k].buf;
			const chaindp_pkt_hdr_t *h = (const chaindp_pkt_hdr_t*)base;
			const char *q = base + sizeof(chaindp_pkt_hdr_t);
			for (int i = 0; i < (int)h->num; ++i) {
				const chaindp_pkt_task_t *t = (const chaindp_pkt_task_t*)q;
				ReadRef rr;
				rr.task = t; rr.anchors = (const chaindp_anchor_t*)(q + sizeof(chaindp_pkt_task_t));
				rr.mini = nullptr; rr.n_anchors = t->seednum > 0 ? t->seednum : 0; rr.rep_len = 0; rr.n_minipos = 0;
				rr.pkt = (int)k; rr.idx = i; rr.batch_read = -1;
				rr.on_device = h->type == CHAINDP_PKT_ANCHORS && t->seednum >= 0 && t->gap_ref >= 0 && t->gap_qry >= 0;
				// a read that one device batch cannot hold is answered the way the reference's device says "cannot do it":
				// err_flag = 1, the host recomputes it (map.c:933-944)
				if (rr.on_device && rr.n_anchors > cap_anchors) rr.on_device = false;
				if (h->type == CHAINDP_PKT_MINIMIZERS && have_index && t->seednum >= 0 && t->gap_ref >= 0 && t->gap_qry >= 0) {
					rr.mini = rr.anchors; rr.anchors = nullptr; rr.n_anchors = 0;   // seeds are collected below
					rr.on_device = true;
				}
				const uint64_t payload = CHAINDP_ALIGN64((uint64_t)(t->seednum > 0 ? t->seednum : 0) * sizeof(chaindp_anchor_t));
				q += sizeof(chaindp_pkt_task_t) + payload;
				reads.push_back(rr);
			}
		}
		// ---- work list: reads grouped by payload kind and (gap_ref, gap_qry) (one group in practice), each group cut into device
		// batches that fit the context (anchor packets: by their known counts); a batch whose seeds turn out not to fit is halved
		// and retried
		struct Sub { bool from_minimizers; int gap_ref, gap_qry; std::vector<size_t> idx; };
		std::deque<Sub> work;
		{
			std::map<std::tuple<int, int, int>, std::vector<size_t>> groups;
			for (size_t r = 0; r < reads.size(); ++r)
				if (reads[r].on_device) groups[std::make_tuple(reads[r].mini ? 1 : 0, reads[r].task->gap_ref, reads[r].task->gap_qry)].push_back(r);
			for (auto &kv : groups) {
				Sub sb; sb.from_minimizers = std::get<0>(kv.first) != 0; sb.gap_ref = std::get<1>(kv.first); sb.gap_qry = std::get<2>(kv.first);
				int64_t acc = 0;
				for (size_t r : kv.second) {
					const int64_t n = sb.from_minimizers ? 0 : reads[r].n_anchors;
					if (!sb.idx.empty() && (acc + n > cap_anchors || (int64_t)sb.idx.size() >= cap_reads)) { work.push_back(sb); sb.idx.clear(); acc = 0; }
					sb.idx.push_back(r); acc += n;
				}
				if (!sb.idx.empty()) work.push_back(sb);
			}
		}
		const bool direct = work.size() <= 1;      // one device batch: the device writes straight into the result packets
		std::vector<int64_t> n_a(reads.size(), 0);
		std::vector<const chaindp_seed_t*> seed_src(reads.size(), nullptr);   // staged new_seed[] (several device batches)
		std::vector<void*> group_stage;
		std::vector<Result> out;
		std::vector<chaindp_seed_t*> seed_dst(reads.size(), nullptr);         // where each read's records go in its result packet
		std::vector<uint64_t*> minipos_dst(reads.size(), nullptr);            // ... and its mini_pos[]
		int64_t n_reads_done = 0, n_anchors_done = 0, n_err = 0, n_batches = 0;
		bool built = false;

		// lays out the result packets (map.c:494-567 writes them the same way; parsed at map.c:918-931):
		// headers are written by the host, the new_seed[] payload either by the device (one device batch: the
		// scatter kernel writes straight into the pinned packet) or copied from a staging buffer.
		auto build_packets = [&]() {
			size_t r0 = 0;
			for (size_t k = 0; k < pk.size(); ++k) {
				const chaindp_pkt_hdr_t *h = (const chaindp_pkt_hdr_t*)pk[k].buf;
				size_t bytes = sizeof(chaindp_pkt_hdr_t);
				for (int i = 0; i < (int)h->num; ++i) {
					bytes += sizeof(chaindp_pkt_result_t);
					if (reads[r0 + i].on_device) bytes += CHAINDP_ALIGN64((uint64_t)n_a[r0 + i] * sizeof(chaindp_seed_t))
					                                       + CHAINDP_ALIGN64((uint64_t)reads[r0 + i].n_minipos * sizeof(uint64_t));
				}
				char *ob = (char*)g.pool.get(bytes);
				if (!ob) fail_hard("out of pinned memory");
				chaindp_pkt_hdr_t *oh = (chaindp_pkt_hdr_t*)ob;
				memset(oh, 0, sizeof(*oh));
				oh->magic = h->magic; oh->size = (uint32_t)bytes; oh->tid = h->tid; oh->num = h->num; oh->type = h->type; oh->lat = h->lat;
				char *q = ob + sizeof(chaindp_pkt_hdr_t);
				for (int i = 0; i < (int)h->num; ++i) {
					const ReadRef &rr = reads[r0 + i];
					chaindp_pkt_result_t *res = (chaindp_pkt_result_t*)q;
					memset(res, 0, sizeof(*res));
					res->read_id = rr.task->read_id;
					q += sizeof(chaindp_pkt_result_t);
					if (!rr.on_device) {
						res->err_flag = 1; res->sub_size = sizeof(chaindp_pkt_result_t);   // header only (map.c:970-971)
						++n_err;
					} else {
						const uint64_t sbytes = (uint64_t)n_a[r0 + i] * sizeof(chaindp_seed_t), sp = CHAINDP_ALIGN64(sbytes);
						const uint64_t mb = (uint64_t)rr.n_minipos * sizeof(uint64_t), mp = CHAINDP_ALIGN64(mb);
						res->n_a = (uint32_t)n_a[r0 + i];
						res->n_minipos = (uint32_t)rr.n_minipos; res->rep_len = rr.rep_len;          // map.c:530-531; zero for anchor packets
						res->sub_size = (uint32_t)(sizeof(chaindp_pkt_result_t) + sp + mp);
						seed_dst[r0 + i] = (chaindp_seed_t*)q;
						if (seed_src[r0 + i]) {
							if (sbytes) memcpy(q, seed_src[r0 + i], sbytes);
							if (sp > sbytes) memset(q + sbytes, 0, sp - sbytes);
						}
						q += sp;
						if (mp) {                                                             // mini_pos[] behind new_seed[] (map.c:547-552)
							minipos_dst[r0 + i] = (uint64_t*)q;                               // written by the device, or from the staged copy
							if (!rr.mini_pos.empty()) { memcpy(q, rr.mini_pos.data(), mb); if (mp > mb) memset(q + mb, 0, mp - mb); }
							q += mp;
						}
						n_anchors_done += rr.n_anchors;
					}
					++n_reads_done;
				}
				r0 += h->num;
				out.push_back(Result{ob, (int)bytes});
			}
		};

		static const bool trace = getenv("CHAINDP_SHIM_TRACE") != nullptr;       // (read once: getenv is not safe against a concurrent setenv)
		auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		while (!work.empty()) {
			const Sub sb = work.front();
			work.pop_front();
			const std::vector<size_t> &idx = sb.idx;
			chaindp_params_t par;
			par.max_dist_x = sb.gap_ref; par.max_dist_y = sb.gap_qry; par.bw = bw; par.max_skip = max_skip;
			par.min_sc = min_sc; par.is_cdna = is_cdna; par.n_segs = 1;
			std::vector<int64_t> off(idx.size() + 1, 0);
			std::vector<const chaindp_anchor_t*> ptrs(idx.size());
			std::vector<int32_t> nseg(idx.size());
			for (size_t k = 0; k < idx.size(); ++k) {
				const ReadRef &rr = reads[idx[k]];
				off[k + 1] = off[k] + rr.n_anchors;
				ptrs[k] = rr.anchors;
				nseg[k] = rr.task->n_segs;
			}
			std::vector<int64_t> soff(idx.size() + 1, 0);
			int rc;
			const double t_a = tnow();
			double t_b = t_a, t_c = t_a;
			if (sb.from_minimizers) {
				// the reference's device did the seed lookup (map.c:523 inside fpga_work): the read's minimizers go to the GPU,
				// which looks them up in the index image, expands, sorts (chaindp_seed.hip) and leaves the anchors in HBM
				std::vector<int64_t> moff(idx.size() + 1, 0), mpoff(idx.size() + 1, 0);
				std::vector<uint32_t> bids(idx.size());
				std::vector<int32_t> qlens(idx.size()), rlen(idx.size());
				std::vector<const chaindp_anchor_t*> mptr(idx.size());
				for (size_t k = 0; k < idx.size(); ++k) {
					const ReadRef &rr = reads[idx[k]];
					moff[k + 1] = moff[k] + rr.task->seednum;
					mptr[k] = rr.mini; bids[k] = rr.task->bid; qlens[k] = rr.task->qlensum;
				}
				t_b = tnow();
				rc = chaindp_collect_seeds_gather(ctx, dev_index, sflag, max_occ, (int64_t)idx.size(), moff.data(), mptr.data(), bids.data(), qlens.data(),
				                                  nseg.data(), off.data(), rlen.data(), mpoff.data());
				t_c = tnow();
				if (rc == CHAINDP_ERR_CAPACITY) {
					// more seeds than one device batch holds (repeat-rich input: tens of hits per minimizer).  Halve the read set and try
					// again; only a read that does not fit on its own is handed back (err_flag = 1, map.c:933-944).
					bool warn = false;
					{ std::lock_guard<std::mutex> lk(g.mu); warn = !g.warned_capacity; g.warned_capacity = true; }
					if (warn) fprintf(stderr, "[chaindp-fpga] a batch of %zu reads has more seeds than a device batch holds (%lld anchors): splitting it; "
					                          "reads that do not fit alone go back to the host with err_flag = 1 (this message is printed once)\n",
					                  idx.size(), (long long)cap_anchors);
					if (idx.size() == 1) { reads[idx[0]].on_device = false; reads[idx[0]].mini_pos.clear(); reads[idx[0]].n_minipos = 0; }
					else {
						Sub lo = sb, hi = sb;
						lo.idx.assign(idx.begin(), idx.begin() + idx.size() / 2);
						hi.idx.assign(idx.begin() + idx.size() / 2, idx.end());
						work.push_front(hi); work.push_front(lo);
					}
					continue;
				}
				if (rc == CHAINDP_OK) for (size_t k = 0; k < idx.size(); ++k) {
					ReadRef &rr = reads[idx[k]];
					rr.n_anchors = off[k + 1] - off[k]; rr.rep_len = rlen[k]; rr.n_minipos = mpoff[k + 1] - mpoff[k];
				}
				if (rc == CHAINDP_OK && !direct) {                                        // several device batches: packets are assembled from staged copies
					std::vector<uint64_t> mp((size_t)mpoff[idx.size()] + 1);
					rc = chaindp_download_mini_pos(ctx, mp.data());
					if (rc == CHAINDP_OK) for (size_t k = 0; k < idx.size(); ++k)
						reads[idx[k]].mini_pos.assign(mp.begin() + mpoff[k], mp.begin() + mpoff[k + 1]);
				}
			} else {
				// the packets are pinned driver buffers: one gather kernel pulls every read's anchors over PCIe
				rc = chaindp_upload_gather_ex(ctx, (int64_t)idx.size(), off.data(), ptrs.data(), nseg.data(), 1);
			}
			const double t_d = tnow();
			if (rc == CHAINDP_OK) rc = chaindp_run(ctx, &par);
			if (rc == CHAINDP_OK) rc = chaindp_compact_offsets(ctx, &par, soff.data());
			const double t_e = tnow();
			if (rc != CHAINDP_OK) { fprintf(stderr, "[chaindp-fpga] %s\n", chaindp_last_error(ctx)); fail_hard("device batch failed"); }
			for (size_t k = 0; k < idx.size(); ++k) n_a[idx[k]] = soff[k + 1] - soff[k];
			// `direct` was decided before any split: a batch that was split after all takes the staged path
			if (direct && work.empty() && !built) {
				build_packets(); built = true;
				std::vector<chaindp_seed_t*> dst(idx.size());
				for (size_t k = 0; k < idx.size(); ++k) dst[k] = seed_dst[idx[k]];
				rc = chaindp_scatter_seeds(ctx, (int64_t)idx.size(), dst.data());      // device writes into the result packets
				if (rc == CHAINDP_OK && sb.from_minimizers) {
					std::vector<uint64_t*> mdst(idx.size());
					for (size_t k = 0; k < idx.size(); ++k) mdst[k] = minipos_dst[idx[k]];
					rc = chaindp_scatter_mini_pos(ctx, (int64_t)idx.size(), mdst.data());
				}
				if (rc == CHAINDP_OK) rc = chaindp_sync(ctx);
			} else {
				const int64_t m = soff[idx.size()];
				void *stage = g.pool.get((size_t)(m > 0 ? m : 1) * sizeof(chaindp_seed_t));
				if (!stage) fail_hard("out of pinned memory");
				rc = chaindp_download_seeds(ctx, 0, m, (chaindp_seed_t*)stage);
				if (rc == CHAINDP_OK && sb.from_minimizers && direct) {               // (a split batch: its mini_pos was not staged above)
					std::vector<int64_t> mpo(idx.size() + 1, 0);
					for (size_t k = 0; k < idx.size(); ++k) mpo[k + 1] = mpo[k] + reads[idx[k]].n_minipos;
					std::vector<uint64_t> mp((size_t)mpo[idx.size()] + 1);
					rc = chaindp_download_mini_pos(ctx, mp.data());
					if (rc == CHAINDP_OK) for (size_t k = 0; k < idx.size(); ++k)
						reads[idx[k]].mini_pos.assign(mp.begin() + mpo[k], mp.begin() + mpo[k + 1]);
				}
				if (rc == CHAINDP_OK) rc = chaindp_sync(ctx);
				for (size_t k = 0; k < idx.size(); ++k) seed_src[idx[k]] = (const chaindp_seed_t*)stage + soff[k];
				group_stage.push_back(stage);
			}
			if (rc != CHAINDP_OK) { fprintf(stderr, "[chaindp-fpga] %s\n", chaindp_last_error(ctx)); fail_hard("device batch failed"); }
			if (trace) fprintf(stderr, "[chaindp-fpga] gpu %d: batch of %zu reads, %lld anchors: gather %.2f ms, collect_seeds %.2f, mini_pos+bookkeeping %.2f, run+compact %.2f, packets+scatter %.2f\n",
			                   device, idx.size(), (long long)off[idx.size()], t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, tnow() - t_e);
			++n_batches;
		}
		if (!built) build_packets();
		for (void *st : group_stage) g.pool.put(st);
		{
			std::lock_guard<std::mutex> lk(g.mu);
			for (size_t k = 0; k < pk.size(); ++k) { g.pool.put(pk[k].buf); g.inflight_bytes -= pk[k].granted < g.inflight_bytes ? pk[k].granted : g.inflight_bytes; }
			for (auto &r : out) g.result_q.push_back(r);
			g.stats[0] += (int64_t)pk.size(); g.stats[1] += n_reads_done; g.stats[2] += n_anchors_done; g.stats[3] += n_batches; g.stats[4] += n_err;
			share.batches += n_batches; share.anchors += n_anchors_done;
		}
		g.cv_result.notify_all();
	}
	chaindp_destroy(ctx);
}

} // namespace

extern "C" void chaindp_fpga_configure(int n_gpus, int max_packets_per_batch, unsigned long max_inflight_bytes)
{
	std::lock_guard<std::mutex> lk(g.mu);
	g.n_gpus_cfg = n_gpus;
	if (max_packets_per_batch > 0) g.max_packets = max_packets_per_batch;
	if (max_inflight_bytes > 0) g.max_inflight = max_inflight_bytes;
}

extern "C" void chaindp_fpga_configure_groups(int n_groups)
{
	std::lock_guard<std::mutex> lk(g.mu);
	g.n_groups_cfg = n_groups > 0 ? n_groups : 0;
}

extern "C" void chaindp_fpga_configure_services(int contexts_per_group)
{
	std::lock_guard<std::mutex> lk(g.mu);
	if (contexts_per_group >= 1 && contexts_per_group <= 8) g.services_per_gpu = contexts_per_group;
}

extern "C" void chaindp_fpga_configure_capacity(int64_t max_anchors_per_batch, int64_t max_reads_per_batch)
{
	std::lock_guard<std::mutex> lk(g.mu);
	if (max_anchors_per_batch > 0) g.cap_anchors = max_anchors_per_batch;
	if (max_reads_per_batch > 0) g.cap_reads = max_reads_per_batch;
}

extern "C" void chaindp_fpga_stats(int64_t st[5])
{
	std::lock_guard<std::mutex> lk(g.mu);
	for (int k = 0; k < 5; ++k) st[k] = g.stats[k];
}

extern "C" int chaindp_fpga_stats_gpu(int gpu, int64_t st[2])
{
	std::lock_guard<std::mutex> lk(g.mu);
	if (gpu < 0 || (size_t)gpu >= g.gpus.size() || !st) return -1;
	st[0] = g.gpus[(size_t)gpu]->batches; st[1] = g.gpus[(size_t)gpu]->anchors;
	return (int)g.gpus.size();
}

extern "C" int fpga_init(int flag)
{
	(void)flag;   // BLOCK / NOBLOCK (fpga.h:11-12): the receive side is always blocking until fpga_exit_block
	std::lock_guard<std::mutex> lk(g.mu);
	if (g.up) return 0;
	int n = chaindp_device_count();
	if (n <= 0) {
		fprintf(stderr, "[chaindp-fpga] fpga_init: no MI355X visible to this process; there is no CPU fallback in this library\n");
		return -1;
	}
	if (g.n_gpus_cfg > 0 && g.n_gpus_cfg < n) n = g.n_gpus_cfg;
	g.stopping = false; g.exit_block = false; g.warned_capacity = false;
	for (int k = 0; k < 5; ++k) g.stats[k] = 0;
	// service groups: one per GPU; chaindp_fpga_configure_groups can ask for more than there are GPUs, group k then runs on GPU k mod n
	// with contexts, index copy and counters of its own (how the multi-GPU dispatch is rehearsed on a one-GPU box)
	const int groups = g.n_groups_cfg > 0 ? g.n_groups_cfg : n;
	g.gpus.clear(); g.group_device.clear();
	for (int k = 0; k < groups; ++k) { g.gpus.emplace_back(new GpuShare()); g.group_device.push_back(k % n); }
	// two service threads (two contexts, two streams) per GPU: while one batch is in its kernels the other one's
	// packets cross PCIe, in either direction.  Every thread takes work when it is free, so a GPU that is busy takes none:
	// the GPUs of a node share the packet stream by the work they get done (chaindp_fpga_stats_gpu shows the split).
	for (int d = 0; d < groups; ++d) for (int k = 0; k < g.services_per_gpu; ++k) g.workers.emplace_back(service_loop, d);
	g.up = true;
	return 0;
}

extern "C" void fpga_finalize(void)
{
	{
		std::lock_guard<std::mutex> lk(g.mu);
		g.building.reset(); g.sealed.reset();  // the index image belongs to the session that loaded it
		if (!g.up) return;
		g.stopping = true;
	}
	g.cv_submit.notify_all();
	for (auto &t : g.workers) t.join();
	g.workers.clear();
	{
		std::lock_guard<std::mutex> lk(g.mu);
		g.submit_q.clear(); g.result_q.clear(); g.inflight_bytes = 0;
		for (auto &sh : g.gpus) sh->idx.reset();         // (the workers are joined: no batch holds a reference any more)
		g.up = false;
	}
	g.pool.destroy();
}

extern "C" void fpga_set_params(int bw, int is_cdna, int max_skip, int min_sc, int flag, int max_occ)
{
	std::lock_guard<std::mutex> lk(g.mu);
	g.bw = bw; g.is_cdna = is_cdna; g.max_skip = max_skip; g.min_sc = min_sc; g.flag = flag; g.max_occ = max_occ;
	seal_index_locked();                       // main.c:243 follows the index part's fpga_load_index calls (main.c:201-204)
}

extern "C" void fpga_load_index(void *addr, int size, int type)
{
	// index.c:102-119 streams the B/H/V/P index image (types 4..7) to the FPGA, which did the seed lookup itself.
	// Here the image is kept until the first service context of each GPU copies it to HBM, where the lookup runs
	// (chaindp_seed.hip) for packets that carry minimizers (type 3, the unmodified reference); anchor packets (type 0x41)
	// do not need it.  The image in use is never modified: chunks go to a new one, which replaces it when it is sealed.
	if (!addr || size <= 0) return;
	std::lock_guard<std::mutex> lk(g.mu);
	if (!g.building) g.building = std::make_shared<IndexImage>();
	g.building->append(type, addr, (size_t)size);
}

extern "C" void *fpga_get_writebuf_thread(unsigned long size, int type, int tid)
{
	(void)type; (void)tid;
	{
		std::lock_guard<std::mutex> lk(g.mu);
		if (!g.up) fail_hard("fpga_get_writebuf_thread before a successful fpga_init (a NULL here would make the caller retry forever, map.c:439)");
		if (g.inflight_bytes + size > g.max_inflight && g.inflight_bytes > 0) return nullptr;   // busy: caller usleep(50)s and retries
		g.inflight_bytes += size;              // released by the size GRANTED here when the packet's batch completes
	}
	void *p = g.pool.get(size);
	if (!p) {
		std::lock_guard<std::mutex> lk(g.mu);
		g.inflight_bytes -= size < g.inflight_bytes ? size : g.inflight_bytes;
	}
	return p;
}

extern "C" void *fpga_get_writebuf(unsigned long size, int type)
{
	{
		std::lock_guard<std::mutex> lk(g.mu);
		if (!g.up) {
			fprintf(stderr, "[chaindp-fpga] fpga_get_writebuf before a successful fpga_init\n");
			return nullptr;                       // the reference exit(1)s on NULL here (fpga_chaindp.c:105-109)
		}
	}
	return fpga_get_writebuf_thread(size, type, -1);
}

extern "C" int fpga_writebuf_submit(void *addr, unsigned int size, unsigned int type)
{
	(void)type;   // callers pass TYPE_CD while the header says 3 (map.c:302,444); the header decides
	if (!addr || size < sizeof(chaindp_pkt_hdr_t)) return -1;
	const size_t granted = g.pool.granted(addr);
	if (granted == 0 || size > granted) return -1;                  // not one of ours, or more than the buffer was asked for
	// the task headers and payloads must lie inside what the caller says it filled in (map.c:484-568 trusts them)
	{
		const chaindp_pkt_hdr_t *h = (const chaindp_pkt_hdr_t*)addr;
		uint64_t pos = sizeof(chaindp_pkt_hdr_t);
		for (int i = 0; i < (int)h->num; ++i) {
			if (pos + sizeof(chaindp_pkt_task_t) > size) return -1;
			const chaindp_pkt_task_t *t = (const chaindp_pkt_task_t*)((const char*)addr + pos);
			pos += sizeof(chaindp_pkt_task_t) + CHAINDP_ALIGN64((uint64_t)(t->seednum > 0 ? t->seednum : 0) * sizeof(chaindp_anchor_t));
			if (pos > size) return -1;
		}
	}
	{
		std::lock_guard<std::mutex> lk(g.mu);
		if (!g.up) return -1;
		seal_index_locked();                   // a packet closes the image that was being received
		chaindp_pkt_hdr_t *h = (chaindp_pkt_hdr_t*)addr;
		if (h->magic == 0) h->magic = g.next_magic++;      // send_task_thread numbered packets (fpga_chaindp.c:99-100)
		g.submit_q.push_back(Submitted{addr, size, granted});
	}
	g.cv_submit.notify_one();
	return 0;
}

extern "C" void *fpga_get_retbuf(int *len, int type)
{
	(void)type;
	std::unique_lock<std::mutex> lk(g.mu);
	g.cv_result.wait(lk, [] { return !g.result_q.empty() || g.exit_block || !g.up; });
	if (g.result_q.empty()) { if (len) *len = 0; return nullptr; }   // fpga_exit_block: len == 0 ends recv_task_thread (fpga_chaindp.c:242)
	Result r = g.result_q.front();
	g.result_q.pop_front();
	if (len) *len = r.size;
	return r.buf;
}

extern "C" int fpga_release_retbuf(void *addr)
{
	if (!addr) return -1;
	return g.pool.put(addr) ? 0 : -1;          // (a second release of the same buffer is refused)
}

extern "C" void fpga_exit_block(void)
{
	{ std::lock_guard<std::mutex> lk(g.mu); g.exit_block = true; }
	g.cv_result.notify_all();
}

extern "C" void fpga_set_block(void)
{
	std::lock_guard<std::mutex> lk(g.mu);
	g.exit_block = false;
}
