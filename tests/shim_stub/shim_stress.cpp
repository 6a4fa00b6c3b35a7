// TEST INFRASTRUCTURE ONLY -- drives csrc/fpga_shim.cpp (built against chaindp_stub.cpp) the way the reference does
// (producers: map.c:439-444, one receiver: fpga_chaindp.c:228-266) from several threads, with both packet kinds, an index image
// that is replaced mid-stream, a byte budget small enough for the busy/NULL path, and a batch capacity small enough for the
// split-and-retry and err_flag paths.  Every result packet is checked; run under -fsanitize=thread and -fsanitize=address.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/chaindp_fpga.h"
extern "C" int chaindp_stub_live_indexes(void);
extern "C" int chaindp_stub_index_creates(void);

static std::atomic<long> n_results{0}, n_reads_ok{0}, n_reads_err{0}, n_bad{0}, n_busy{0};
static const int PRODUCERS = 6, PACKETS_EACH = 60, READS_PER_PACKET = 8, RESEAL_EVERY = 10;

static uint32_t rng(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
static int payload_len(uint32_t read_id) { uint32_t s = read_id * 2654435761u + 1; return (int)(rng(s) % 300); }   // anchors / minimizers of the read
static chaindp_anchor_t element(uint32_t read_id, int k) { chaindp_anchor_t a; a.x = (uint64_t)read_id << 32 | (uint32_t)k; a.y = (uint64_t)(k * 7 + read_id); return a; }

static void producer(int tid)
{
	for (int pk = 0; pk < PACKETS_EACH; ++pk) {
		const int type = (pk + tid) % 3 == 0 ? CHAINDP_PKT_MINIMIZERS : CHAINDP_PKT_ANCHORS;
		size_t bytes = 64;
		uint32_t ids[READS_PER_PACKET];
		for (int i = 0; i < READS_PER_PACKET; ++i) { ids[i] = (uint32_t)((tid * PACKETS_EACH + pk) * READS_PER_PACKET + i); bytes += 64 + CHAINDP_ALIGN64((uint64_t)payload_len(ids[i]) * 16); }
		void *buf;
		while (!(buf = fpga_get_writebuf_thread(bytes, 0, tid))) { ++n_busy; usleep(50); }     // map.c:439-441
		chaindp_pkt_hdr_t *h = (chaindp_pkt_hdr_t*)buf;
		memset(h, 0, 64); h->size = (uint32_t)bytes; h->tid = (uint16_t)tid; h->num = READS_PER_PACKET; h->type = (uint8_t)type;
		char *q = (char*)buf + 64;
		for (int i = 0; i < READS_PER_PACKET; ++i) {
			chaindp_pkt_task_t *t = (chaindp_pkt_task_t*)q;
			memset(t, 0, 64);
			const int n = payload_len(ids[i]);
			t->gap_qry = 5000; t->gap_ref = (ids[i] % 5 == 0) ? 4000 : 5000; t->seednum = n; t->read_id = ids[i]; t->n_segs = 1;
			chaindp_anchor_t *pl = (chaindp_anchor_t*)(q + 64);
			for (int k = 0; k < n; ++k) pl[k] = element(ids[i], k);
			const size_t pb = CHAINDP_ALIGN64((uint64_t)n * 16);
			if (pb > (size_t)n * 16) memset((char*)pl + (size_t)n * 16, 0, pb - (size_t)n * 16);
			q += 64 + pb;
		}
		if (fpga_writebuf_submit(buf, (unsigned)bytes, 1) != 0) { ++n_bad; }
		if (tid == 0 && pk % RESEAL_EVERY == RESEAL_EVERY - 1) {           // a new index part in mid-stream (main.c:201-204, 243), with
			std::vector<uint8_t> blob(4096, 7);                                // both contexts of both GPUs holding batches in flight
			for (int t = 4; t <= 7; ++t) fpga_load_index(blob.data(), (int)blob.size(), t);
			fpga_set_params(500, 0, 25, 40, 0, 100);
		}
	}
}

static void receiver(long expect_packets, int hits, long cap)
{
	long got = 0;
	while (got < expect_packets) {
		int len = 0;
		void *p = fpga_get_retbuf(&len, 3);
		if (len == 0) break;
		std::vector<char> copy((char*)p, (char*)p + len);                     // fpga_chaindp.c:250-262: copy, then release
		if (fpga_release_retbuf(p) != 0) ++n_bad;
		const chaindp_pkt_hdr_t *h = (const chaindp_pkt_hdr_t*)copy.data();
		if ((int)h->size != len || h->num != READS_PER_PACKET) ++n_bad;
		const char *q = copy.data() + 64;
		for (int i = 0; i < h->num; ++i) {
			const chaindp_pkt_result_t *r = (const chaindp_pkt_result_t*)q;
			const int n = payload_len(r->read_id);
			const long seeds = h->type == CHAINDP_PKT_MINIMIZERS ? (long)n * hits : n;
			if (r->err_flag) {
				if (r->sub_size != 64 || seeds <= cap) ++n_bad;                  // only a read that does not fit alone may come back like this
				++n_reads_err; q += 64; continue;
			}
			if ((long)r->n_a != seeds) ++n_bad;
			const chaindp_seed_t *s = (const chaindp_seed_t*)(q + 64);
			for (long k = 0; k < (long)r->n_a; ++k) {
				const chaindp_anchor_t e = element(r->read_id, (int)(h->type == CHAINDP_PKT_MINIMIZERS ? k / hits : k));
				if (s[k].seed.x != e.x || s[k].seed.y != e.y || s[k].p != -4) { ++n_bad; break; }
			}
			if (h->type == CHAINDP_PKT_MINIMIZERS && (int)r->n_minipos != n) ++n_bad;
			++n_reads_ok;
			q += r->sub_size;
		}
		if (q - copy.data() != len) ++n_bad;
		++got; ++n_results;
	}
}

int main(int argc, char **argv)
{
	const int hits = argc > 1 ? atoi(argv[1]) : 1;
	const long cap = argc > 2 ? atol(argv[2]) : (32l << 20);
	char hv[32]; snprintf(hv, sizeof hv, "%d", hits); setenv("CHAINDP_STUB_HITS", hv, 1);
	chaindp_fpga_configure(2, 5, 1ul << 20);                  // two "GPUs", small batches, 1 MiB in flight: producers see NULL often
	chaindp_fpga_configure_capacity(cap, 1 << 19);
	if (fpga_init(0) != 0) { fprintf(stderr, "fpga_init failed\n"); return 2; }
	std::vector<uint8_t> blob(8192, 3);
	for (int t = 4; t <= 7; ++t) fpga_load_index(blob.data(), (int)blob.size(), t);
	fpga_set_params(500, 0, 25, 40, 0, 100);
	std::thread rx(receiver, (long)PRODUCERS * PACKETS_EACH, hits, cap);
	std::vector<std::thread> tx;
	for (int t = 0; t < PRODUCERS; ++t) tx.emplace_back(producer, t);
	for (auto &t : tx) t.join();
	rx.join();
	int64_t st[5], sg[2]; chaindp_fpga_stats(st);
	const int ngpu = chaindp_fpga_stats_gpu(0, sg);
	fpga_exit_block(); fpga_set_block(); fpga_finalize();
	printf("packets %ld reads ok %ld err %ld bad %ld busy %ld batches %lld gpus %d index_creates %d live_indexes %d\n", n_results.load(), n_reads_ok.load(),
	       n_reads_err.load(), n_bad.load(), n_busy.load(), (long long)st[3], ngpu, chaindp_stub_index_creates(), chaindp_stub_live_indexes());
	const bool ok = n_bad == 0 && n_results == PRODUCERS * PACKETS_EACH && n_reads_ok + n_reads_err == (long)PRODUCERS * PACKETS_EACH * READS_PER_PACKET &&
	                st[4] == n_reads_err && chaindp_stub_live_indexes() == 0 &&
	                chaindp_stub_index_creates() <= 2 * 2 * (1 + PACKETS_EACH / RESEAL_EVERY);   // one copy per GPU and image, plus at most one private
	                                                                                         // copy per image for a context that still held the older one
	return ok ? 0 : 1;
}
