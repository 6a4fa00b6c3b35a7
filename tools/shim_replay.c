/*
 * shim_replay.c -- a C host that replays a file of ready-made task packets through the reference's driver ABI
 * (fpga.h:37-62), the way the reference's threads do (map.c:439-444 producers, fpga_chaindp.c:228-270 receiver),
 * after streaming an index image with fpga_load_index (main.c:201-204).  Used for the reference's own minimizer
 * packets (type 3): tools/shim_minimizer_bench.py --write-replay makes the file from a seed dump.  Prints the
 * PCIe-inclusive anchors/s (anchors = new_seed candidates the device collected: sum of seed counts is reported too).
 *   gcc -O2 -o tools/shim_replay tools/shim_replay.c -Iinclude -Lminimap2_chaindp_amd/csrc -lchaindp_hip -lpthread
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
	if (argc < 2) { fprintf(stderr, "usage: shim_replay <file> [producers=8] [reps=4] [rounds=3]\n"); return 2; }
	n_prod = argc > 2 ? atoi(argv[2]) : 8;
	reps = argc > 3 ? atoi(argv[3]) : 4;
	int rounds = argc > 4 ? atoi(argv[4]) : 3, k, round;
	FILE *fp = fopen(argv[1], "rb");
	char magic[8];
	int32_t hdr[6];
	if (!fp || fread(magic, 1, 8, fp) != 8 || memcmp(magic, "SHIMRPL1", 8) || fread(hdr, 4, 6, fp) != 6) { fprintf(stderr, "bad replay file\n"); return 1; }
	n_packets = hdr[5];
	chaindp_fpga_configure(0, 256, 0);
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
	int64_t minimizers = 0;
	for (k = 0; k < n_packets; ++k) {
		if (fread(&pkt_size[k], 4, 1, fp) != 1) return 1;
		pkt[k] = (char*)malloc(pkt_size[k]);
		if (fread(pkt[k], 1, pkt_size[k], fp) != pkt_size[k]) return 1;
		minimizers += (pkt_size[k] - 64) / 16;
	}
	fclose(fp);
	pthread_t rx, *tx = (pthread_t*)malloc(sizeof(pthread_t) * n_prod);
	pthread_create(&rx, 0, receiver, 0);
	for (round = 0; round < rounds; ++round) {
		const int64_t want = got_packets + (int64_t)n_packets * reps, seeds0 = got_seeds;
		double t0 = now();
		for (k = 0; k < n_prod; ++k) pthread_create(&tx[k], 0, producer, (void*)(intptr_t)k);
		for (k = 0; k < n_prod; ++k) pthread_join(tx[k], 0);
		while (got_packets < want) usleep(200);
		double dt = now() - t0;
		printf("round %d: %d packets x %d, ~%.1f M minimizers -> %lld new_seed records in %.1f ms: %.1f M records/s, %.1f M minimizers/s (err reads so far %lld)\n",
		       round, n_packets, reps, minimizers * reps / 1e6, (long long)(got_seeds - seeds0), dt * 1e3, (got_seeds - seeds0) / dt / 1e6,
		       minimizers * reps / dt / 1e6, (long long)got_err);
		fflush(stdout);
	}
	fpga_exit_block();                                                                               /* main.c:608 */
	pthread_join(rx, 0);
	fpga_set_block(); fpga_finalize();                                                               /* main.c:613-614 */
	return 0;
}
