/*
 * shim_client.c -- a C host that drives the reference's driver ABI (fpga.h:37-62) the way the reference
 * does: P producer threads build 8-read anchor packets (map.c:286-324), obtain a driver buffer with
 * fpga_get_writebuf_thread (retrying on NULL, map.c:439-441), memcpy the packet in (map.c:442) and submit
 * it (map.c:444); one receiver thread plays recv_task_thread (fpga_chaindp.c:228-270): fpga_get_retbuf,
 * malloc+memcpy, fpga_release_retbuf.  Anchors come from libanchorgen.  Prints PCIe-inclusive anchors/s.
 *   gcc -O2 -o shim_client tools/shim_client.c -Iinclude -Lminimap2_chaindp_amd/csrc -lchaindp_hip -lanchorgen -lpthread
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "chaindp_fpga.h"

typedef struct { int32_t v[17]; } ag_config_t;
int64_t ag_offsets(const ag_config_t*, uint64_t, int64_t, int64_t, int64_t*, int);
void ag_fill(const ag_config_t*, uint64_t, int64_t, int64_t, const int64_t*, void*, int);

static int64_t n_reads, *off;
static chaindp_anchor_t *anchors;
static int n_prod, per_packet = 8;
static volatile int64_t got_packets, got_reads, got_seeds;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void *producer(void *arg)
{
	int tid = (int)(intptr_t)arg;
	int64_t n_pk = (n_reads + per_packet - 1) / per_packet, k;
	for (k = tid; k < n_pk; k += n_prod) {
		int64_t r0 = k * per_packet, r1 = r0 + per_packet < n_reads ? r0 + per_packet : n_reads, r;
		size_t size = sizeof(chaindp_pkt_hdr_t);
		for (r = r0; r < r1; ++r) size += sizeof(chaindp_pkt_task_t) + CHAINDP_ALIGN64((uint64_t)(off[r + 1] - off[r]) * 16);
		char *buf;
		while ((buf = (char*)fpga_get_writebuf_thread(size, 0, tid)) == NULL) usleep(50);      /* map.c:439-441 */
		chaindp_pkt_hdr_t *h = (chaindp_pkt_hdr_t*)buf;
		memset(h, 0, sizeof(*h));
		h->size = (uint32_t)size; h->tid = (uint16_t)tid; h->num = (uint16_t)(r1 - r0); h->type = CHAINDP_PKT_ANCHORS;
		char *q = buf + sizeof(*h);
		for (r = r0; r < r1; ++r) {
			chaindp_pkt_task_t *t = (chaindp_pkt_task_t*)q;
			int64_t n = off[r + 1] - off[r];
			memset(t, 0, sizeof(*t));
			t->gap_qry = 10000; t->gap_ref = 10000; t->seednum = (int32_t)n; t->read_id = (uint32_t)r; t->n_segs = 1;
			memcpy(q + sizeof(*t), anchors + off[r], (size_t)n * 16);                              /* map.c:311 */
			q += sizeof(*t) + CHAINDP_ALIGN64((uint64_t)n * 16);
		}
		fpga_writebuf_submit(buf, (unsigned)size, 1);                                              /* map.c:444 */
	}
	return 0;
}

static void *receiver(void *arg)
{
	(void)arg;
	for (;;) {
		int len = 0;
		void *p = fpga_get_retbuf(&len, 3);                                                        /* fpga_chaindp.c:241 */
		if (len == 0) return 0;
		char *copy = (char*)malloc(len);                                                           /* fpga_chaindp.c:259-261 */
		memcpy(copy, p, len);
		fpga_release_retbuf(p);
		const chaindp_pkt_hdr_t *h = (const chaindp_pkt_hdr_t*)copy;
		const char *q = copy + sizeof(*h);
		for (int i = 0; i < h->num; ++i) {
			const chaindp_pkt_result_t *res = (const chaindp_pkt_result_t*)q;
			got_seeds += res->n_a;
			q += res->sub_size;
		}
		got_reads += h->num;
		__sync_fetch_and_add(&got_packets, 1);
		free(copy);
	}
}

int main(int argc, char **argv)
{
	n_reads = argc > 1 ? atoll(argv[1]) : 8000;
	n_prod = argc > 2 ? atoi(argv[2]) : 8;
	int reps = argc > 3 ? atoi(argv[3]) : 3;
	ag_config_t cfg = {{10000, 0, 40, 10, 40, 30, 6, 8, 1, 15, 0, 100000, 12048, 1, 0, 0, 0}};   /* the "ava-ont" shape */
	off = (int64_t*)malloc((n_reads + 1) * 8);
	int64_t total = ag_offsets(&cfg, 3, 0, n_reads, off, 16);
	anchors = (chaindp_anchor_t*)malloc((size_t)total * 16);
	ag_fill(&cfg, 3, 0, n_reads, off, anchors, 16);
	chaindp_fpga_configure(0, 256, 2ul << 30);
	if (fpga_init(0) != 0) return 1;                                                               /* main.c:512 */
	fpga_set_params(500, 0, 25, 100, 0, 0);                                                        /* main.c:243 */
	pthread_t rx; pthread_create(&rx, 0, receiver, 0);
	int64_t n_pk = (n_reads + per_packet - 1) / per_packet;
	for (int rep = 0; rep < reps; ++rep) {
		pthread_t th[64];
		got_packets = 0; got_reads = 0; got_seeds = 0;
		double t0 = now();
		for (int t = 0; t < n_prod; ++t) pthread_create(&th[t], 0, producer, (void*)(intptr_t)t);
		for (int t = 0; t < n_prod; ++t) pthread_join(th[t], 0);
		double t1 = now();
		while (got_packets < n_pk) usleep(100);
		double t2 = now();
		int64_t st[5]; chaindp_fpga_stats(st);
		printf("rep %d: %ld reads %ld anchors: submit %.1f ms, all results %.1f ms -> %.1f M anchors/s (seeds %ld, device batches so far %ld)\n",
		       rep, (long)n_reads, (long)total, (t1 - t0) * 1e3, (t2 - t0) * 1e3, total / (t2 - t0) / 1e6, (long)got_seeds, (long)st[3]);
	}
	fpga_exit_block();                                                                             /* main.c:608 */
	pthread_join(rx, 0);
	fpga_set_block(); fpga_finalize();                                                             /* main.c:613-614 */
	return 0;
}
