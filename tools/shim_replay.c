/*
 * shim_replay.c -- a C host that replays a file of ready-made task packets through the reference's driver ABI
 * (fpga.h:37-62), the way the reference's threads do (map.c:439-444 producers, fpga_chaindp.c:228-270 receiver),
 * after streaming an index image with fpga_load_index (main.c:201-204; empty blobs for anchor packets).  Both packet
 * kinds: the reference's own minimizer packets (type 3: tools/shim_minimizer_bench.py --write-replay, bench.py) and this
 * build's anchor packets (type 0x41).  Prints one JSON line per round (PCIe-inclusive seconds, payload elements in,
 * new_seed records out) and a last one with the shim's batch counts and per-GPU shares.
 *   make -C minimap2_chaindp_amd/csrc shim_replay          (bench.py's packet_abi entry runs it)
 *   shim_replay <file> [producers=8] [reps=4] [rounds=3] [max_packets_per_batch=256] [n_gpus=0 (all)] [service contexts per GPU=0 (default)]
 * File: "SHIMRPL1", int32 flag, mid_occ, bw, max_skip, min_sc, n_packets; 4 x (int64 bytes, blob); n_packets x (uint32 size, packet).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "chaindp_fpga.h"

static int n_packets, n_prod, reps;
static char **pkt;
static uint32_t *pkt_size;
static volatile int64_t got_packets, got_reads, got_seeds, got_err;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void *producer(void *arg)
{
	int tid = (int)(intptr_t)arg, rep, k;
	for (rep = 0; rep < reps; ++rep)
		for (k = tid; k < n_packets; k += n_prod) {
			char *buf;
			while ((buf = (char*)fpga_get_writebuf_thread(pkt_size[k], 0, tid)) == NULL) usleep(50);   /* map.c:439-441 */
			memcpy(buf, pkt[k], pkt_size[k]);                                                        /* map.c:442 */
			fpga_writebuf_submit(buf, pkt_size[k], 1);                                               /* map.c:444 */
		}
	return 0;
}

static void *receiver(void *arg)
{
	(void)arg;
	for (;;) {
		int len = 0, i;
		char *p = (char*)fpga_get_retbuf(&len, 3);
		if (len == 0) return 0;                                                                      /* fpga_chaindp.c:242 */
		chaindp_pkt_hdr_t *h = (chaindp_pkt_hdr_t*)p;
		char *q = p + sizeof(*h);
		for (i = 0; i < (int)h->num; ++i) {
			chaindp_pkt_result_t *r = (chaindp_pkt_result_t*)q;
			if (r->err_flag) ++got_err; else got_seeds += r->n_a;
			q += r->sub_size;
			++got_reads;
		}
		fpga_release_retbuf(p);
		++got_packets;
	}
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: shim_replay <file> [producers=8] [reps=4] [rounds=3] [max_packets_per_batch=256] [n_gpus=0] [services=0]\n"); return 2; }
	n_prod = argc > 2 ? atoi(argv[2]) : 8;
	reps = argc > 3 ? atoi(argv[3]) : 4;
	int rounds = argc > 4 ? atoi(argv[4]) : 3, k, round;
	const int max_pk = argc > 5 ? atoi(argv[5]) : 256, n_gpus = argc > 6 ? atoi(argv[6]) : 0, services = argc > 7 ? atoi(argv[7]) : 0;
	FILE *fp = fopen(argv[1], "rb");
	char magic[8];
	int32_t hdr[6];
	if (!fp || fread(magic, 1, 8, fp) != 8 || memcmp(magic, "SHIMRPL1", 8) || fread(hdr, 4, 6, fp) != 6) { fprintf(stderr, "bad replay file\n"); return 1; }
	n_packets = hdr[5];
	chaindp_fpga_configure(n_gpus, max_pk, 0);
	if (services > 0) chaindp_fpga_configure_services(services);
	if (fpga_init(0) != 0) return 1;                                                                 /* main.c:512 */
	for (k = 0; k < 4; ++k) {
		int64_t nb;
		if (fread(&nb, 8, 1, fp) != 1) return 1;
		char *blob = (char*)malloc(nb ? nb : 1);
		if (nb && fread(blob, 1, nb, fp) != (size_t)nb) return 1;
		int64_t o = 0;
		while (o < nb) { int64_t c = nb - o > (1 << 30) ? (1 << 30) : nb - o; fpga_load_index(blob + o, (int)c, 4 + k); o += c; }   /* index.c:102-119 */
		free(blob);
	}
	fpga_set_params(hdr[2], 0, hdr[3], hdr[4], hdr[0], hdr[1]);                                      /* main.c:243 */
	pkt = (char**)malloc(sizeof(char*) * n_packets); pkt_size = (uint32_t*)malloc(4 * n_packets);
	int64_t elements = 0, reads = 0, bytes = 0;
	int type = 0;
	for (k = 0; k < n_packets; ++k) {
		if (fread(&pkt_size[k], 4, 1, fp) != 1) return 1;
		pkt[k] = (char*)malloc(pkt_size[k]);
		if (fread(pkt[k], 1, pkt_size[k], fp) != pkt_size[k]) return 1;
		const chaindp_pkt_hdr_t *h = (const chaindp_pkt_hdr_t*)pkt[k];
		const char *q = pkt[k] + sizeof(*h);
		int i;
		type = h->type;
		for (i = 0; i < (int)h->num; ++i) {                                                          /* the walk of map.c:484-568 */
			const chaindp_pkt_task_t *t = (const chaindp_pkt_task_t*)q;
			const int64_t n = t->seednum > 0 ? t->seednum : 0;
			elements += n; ++reads;
			q += sizeof(*t) + CHAINDP_ALIGN64((uint64_t)n * sizeof(chaindp_anchor_t));
		}
		bytes += pkt_size[k];
	}
	fclose(fp);
	pthread_t rx, *tx = (pthread_t*)malloc(sizeof(pthread_t) * n_prod);
	pthread_create(&rx, 0, receiver, 0);
	for (round = 0; round < rounds; ++round) {
		const int64_t want = got_packets + (int64_t)n_packets * reps, seeds0 = got_seeds;
		double t0 = now();
		for (k = 0; k < n_prod; ++k) pthread_create(&tx[k], 0, producer, (void*)(intptr_t)k);
		for (k = 0; k < n_prod; ++k) pthread_join(tx[k], 0);
		while (got_packets < want) usleep(100);
		double dt = now() - t0;
		printf("{\"round\": %d, \"packet_type\": %d, \"producers\": %d, \"packets\": %lld, \"reads\": %lld, \"elements_in\": %lld, \"bytes_in\": %lld, "
		       "\"records_out\": %lld, \"seconds\": %.6f, \"err_reads_so_far\": %lld}\n",
		       round, type, n_prod, (long long)n_packets * reps, (long long)reads * reps, (long long)elements * reps, (long long)bytes * reps,
		       (long long)(got_seeds - seeds0), dt, (long long)got_err);
		fflush(stdout);
	}
	{
		int64_t st[5], sg[2];
		int ng, d;
		chaindp_fpga_stats(st);
		ng = chaindp_fpga_stats_gpu(0, sg);
		printf("{\"shim\": {\"packets\": %lld, \"reads\": %lld, \"anchors\": %lld, \"device_batches\": %lld, \"err_reads\": %lld, \"gpus\": [",
		       (long long)st[0], (long long)st[1], (long long)st[2], (long long)st[3], (long long)st[4]);
		for (d = 0; d < ng; ++d) { chaindp_fpga_stats_gpu(d, sg); printf("%s{\"gpu\": %d, \"batches\": %lld, \"anchors\": %lld}", d ? ", " : "", d, (long long)sg[0], (long long)sg[1]); }
		printf("]}}\n");
	}
	fpga_exit_block();                                                                               /* main.c:608 */
	pthread_join(rx, 0);
	fpga_set_block(); fpga_finalize();                                                               /* main.c:613-614 */
	return 0;
}
