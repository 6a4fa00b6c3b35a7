/*
 * chaindp_fpga.h -- the reference's accelerator driver ABI, served by an MI355X.
 *
 * The reference links a closed driver library, libfpgadrv.a (reference Makefile:7), through the C
 * API of fpga.h:37-62.  libchaindp_hip.so exports the eleven entry points the reference actually
 * calls (nm -u of its objects), with the same names, argument meaning and error
 * behaviour, so that the reference links against it in place of libfpgadrv.a:
 *
 *   fpga_init            fpga.h:37   main.c:512      open the device(s), start the service threads
 *   fpga_finalize        fpga.h:39   main.c:614
 *   fpga_set_params      fpga.h:58   main.c:243      global DP parameters (bw, is_cdna, max_skip, min_sc)
 *   fpga_load_index      fpga.h:62   index.c:102-119 the B/H/V/P index image, kept and copied to HBM once per GPU: minimizer packets
 *                                                    are looked up in it on the device (an image is used once the next call that is
 *                                                    not fpga_load_index -- fpga_set_params, a submit -- has closed it)
 *   fpga_get_writebuf[_thread] fpga.h:44-45 map.c:439,475; fpga_chaindp.c:104   driver-owned (pinned) packet buffer
 *   fpga_writebuf_submit fpga.h:46   map.c:444,480   hand the filled packet to the device
 *   fpga_get_retbuf      fpga.h:41   fpga_chaindp.c:241   blocking receive of one result packet
 *   fpga_release_retbuf  fpga.h:42   fpga_chaindp.c:262
 *   fpga_exit_block / fpga_set_block  fpga.h:52,54  main.c:608,613   unblock / re-arm the receiver
 *
 * Packet format: reference fpga_chaindp.h:46-87 (64-byte packed headers), assembled at map.c:286-324
 * and parsed at map.c:484-568 (device side) / map.c:918-931 (host side).  Two task payloads are served
 * (SURVEY 8b "Build's boundary"):
 *   type == 3 (the reference's packets, unmodified): payload of each task = seednum x mm128_t MINIMIZERS.  The
 *       FPGA looked the seeds up in its own index image; here the GPU does (collect_seed_hits, map.c:187-236,
 *       over the image received through fpga_load_index: chaindp_collect_seeds), and chains the anchors where
 *       they are.  Without a complete index image every read of such a packet is answered with err_flag = 1 and
 *       no payload, the reference's own "device cannot handle it" signal -- the host then recomputes that
 *       read on the CPU (map.c:933-944).
 *   type == CHAINDP_PKT_ANCHORS : payload of each task = seednum x mm128_t anchors, sorted by x (what
 *       mm_chain_dp_fpga receives, chain.c:218), for hosts that collect seeds themselves.
 * Result packets are exactly the reference's: header echo, then per read collect_result_t +
 * new_seed[n_a] (padded to 64 B) + mini_pos[n_minipos] (padded to 64 B) with rep_len; for anchor packets
 * n_minipos = 0 and rep_len = 0, since those are by-products of seed collection, which that host already has.
 */
#ifndef CHAINDP_FPGA_H
#define CHAINDP_FPGA_H

#include <stdint.h>
#include "chaindp.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CHAINDP_PKT_MINIMIZERS 3      /* reference map.c:302 */
#define CHAINDP_PKT_ANCHORS    0x41

#define CHAINDP_ALIGN64(n) (((n) + 63) & ~(uint64_t)63)   /* ADDR_ALIGN(n, 64), fpga_chaindp.h:18 */

#pragma pack(push, 1)
typedef struct {            /* == chaindp_sndhdr_t, fpga_chaindp.h:79-87 */
	uint32_t magic;
	uint32_t size;          /* whole packet, header included */
	uint16_t tid;
	uint16_t num;           /* reads in the packet */
	uint8_t  type;
	uint8_t  lat;
	uint8_t  reserve1[50];
} chaindp_pkt_hdr_t;

typedef struct {            /* == collect_task_t, fpga_chaindp.h:46-58 */
	int32_t  gap_qry;       /* max_chain_gap_qry -> max_dist_y */
	int32_t  gap_ref;       /* max_chain_gap_ref -> max_dist_x */
	int32_t  seednum;       /* payload element count (anchors for CHAINDP_PKT_ANCHORS) */
	int32_t  qlensum;
	uint32_t read_id;
	uint32_t bid;
	int16_t  n_segs;
	char     b;
	char     reserve1[1];
	uint64_t mv_a;          /* host pointer in the reference; meaningless to the device */
	char     reserve2[28];
} chaindp_pkt_task_t;

typedef struct {            /* == collect_result_t, fpga_chaindp.h:60-68 */
	uint32_t err_flag;      /* 1: no payload follows, host recomputes on the CPU (map.c:933) */
	uint32_t read_id;
	uint32_t sub_size;      /* bytes of this record incl. this header (map.c:554) */
	uint32_t n_a;           /* new_i: number of new_seed records */
	uint32_t n_minipos;
	uint32_t rep_len;
	char     reserve1[40];
} chaindp_pkt_result_t;
#pragma pack(pop)

/* --- the driver ABI (fpga.h:37-62); BUF_TYPE / RET_TYPE are plain ints at this boundary --- */
int   fpga_init(int flag);
void  fpga_finalize(void);
void *fpga_get_retbuf(int *len, int type);
int   fpga_release_retbuf(void *addr);
void *fpga_get_writebuf(unsigned long size, int type);
void *fpga_get_writebuf_thread(unsigned long size, int type, int tid);
int   fpga_writebuf_submit(void *addr, unsigned int size, unsigned int type);
void  fpga_exit_block(void);
void  fpga_set_block(void);
void  fpga_set_params(int bw, int is_cdna, int max_skip, int min_sc, int flag, int max_occ);
void  fpga_load_index(void *addr, int size, int type);

/* --- additions of this build (not in fpga.h) --- */
/* Service configuration, to be called before fpga_init: GPUs to use (0 = all visible), packets
 * merged into one device batch at most, and the in-flight byte budget after which
 * fpga_get_writebuf_thread answers NULL ("busy, retry": map.c:439-441). */
void chaindp_fpga_configure(int n_gpus, int max_packets_per_batch, unsigned long max_inflight_bytes);
/* Service groups, before fpga_init: by default one per GPU in use.  n_groups > GPUs puts group k on GPU k mod GPUs, each group with
 * service contexts, index copy and counters of its own: the node-level dispatch (a packet stream shared by several groups) can be
 * exercised on a box with one GPU.  0 = default. */
void chaindp_fpga_configure_groups(int n_groups);
/* Service contexts (host thread + device context + stream) per group, before fpga_init; default 2, at most 8.  While one context's
 * batch is in its kernels, the others' packets cross PCIe. */
void chaindp_fpga_configure_services(int contexts_per_group);
/* Capacity of one device batch (per service context; defaults 32 Mi anchors, 512 Ki reads), before fpga_init.  A read
 * with more anchors than that -- or, for minimizer packets, more seeds -- is answered with err_flag = 1 (map.c:933-944); a
 * batch whose seeds do not fit is split and retried. */
void chaindp_fpga_configure_capacity(int64_t max_anchors_per_batch, int64_t max_reads_per_batch);
/* Counters since fpga_init: st[0] packets, st[1] reads, st[2] anchors, st[3] device batches,
 * st[4] reads answered err_flag=1. */
void chaindp_fpga_stats(int64_t st[5]);
/* Per GPU (per service group, see above): st[0] device batches, st[1] anchors it has chained.  Returns the number of GPUs (groups)
 * in use, -1 on a bad index.
 * Service threads take packets when they are free, so the node's GPUs split the stream by the work they get done. */
int chaindp_fpga_stats_gpu(int gpu, int64_t st[2]);


#ifdef __cplusplus
}
#endif
#endif
